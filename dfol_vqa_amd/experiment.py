"""Model factory driven by the reference's config YAML (reference: src/gqa_interpreter_experiments.py:81-262,
src/nsvqa/base_experiment.py:43-47).  The YAML keys are the reference's (CONFIG_YAML.md); only the
builders are here — the training/test orchestration of ExperimentBase.run is out of scope."""

import math

import torch
import torch.nn as nn
import yaml

from .gqa_ops import GQAOntology
from .interpreter import BatchGQABoxFeaturizer, BatchGQAInterpreter
from .visual_oracle import CalibrationLSTMCell, ClassifierOracle, EmbeddingLayer, RegularMLP


def load_config(config_file):
    if isinstance(config_file, dict):
        return config_file
    with open(config_file, 'r') as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def build_ontology(config):                                     # gqa_interpreter_experiments.py:83-91
    return GQAOntology(config['attribute_file'], config['class_file'], config['vocabulary_file'], config.get('word_embedding_file'),
                       relation_json_path=config.get('relation_file'), frequency_json_path=config.get('frequency_file'))


def build_neural_modules(config, ontology):                     # gqa_interpreter_experiments.py:107-198 (classifier oracle only)
    fwd_net = bwd_net = out_net = None
    if config.get('activate_attention_transfer'):               # gqa_interpreter_experiments.py:115-138
        output_dim, max_activation = 4, 10.0
        in_dim = config['word_embedding_dim'] + 1 + 17
        fwd_net = CalibrationLSTMCell(in_dim, config['attention_transfer_state_dim'])
        bwd_net = CalibrationLSTMCell(in_dim, config['attention_transfer_state_dim'])
        out_net = nn.Sequential(nn.Linear(2 * config['attention_transfer_state_dim'], output_dim), nn.Sigmoid())
        out_net[0].weight = nn.Parameter(torch.zeros(output_dim, 2 * config['attention_transfer_state_dim']))
        bias = -math.log(max_activation - 1) * torch.ones(output_dim)
        bias[3] = 0
        out_net[0].bias = nn.Parameter(bias)
        if config.get('freeze_attention_network'):
            for net in (fwd_net, bwd_net, out_net):
                net.requires_grad_(False)
    if config['oracle_output_dim'] != 1 or not config['classifier_oracle']:
        raise NotImplementedError("only classifier_oracle with oracle_output_dim == 1 (every shipped config) is built")
    drop = config['dropout']
    featurizer_network = RegularMLP(config['box_features_dim'], config['oracle_input_dim'], config['featurizer_layers_config'], drop)
    attribute_network = RegularMLP(config['oracle_input_dim'] + 4, config['word_embedding_dim'], config['attribute_network_layers_config'], drop)
    concept_num = len(ontology._vocabulary['idx_to_arg'])
    emb_in = config['oracle_input_dim'] + 4 if config['attribute_network_layers_config'] is None else config['word_embedding_dim']
    weights = torch.zeros(concept_num, emb_in)
    torch.nn.init.normal_(weights)
    glove = ontology.get_embeddings(ontology._vocabulary['idx_to_arg'])
    if glove is not None:                                       # rows start as the GloVe vectors of the concept names (:151-153)
        weights[:, :config['word_embedding_dim']] = torch.from_numpy(glove)
    embedding_network = EmbeddingLayer(emb_in, concept_num, drop, weights, torch.zeros(concept_num), config.get('freeze_embedding_bias', False))
    rel_in = config['relation_features_dim'] if 'relation_features_dim' in config else 2 * config['oracle_input_dim'] + 2 * 4 + 4
    relation_network = RegularMLP(rel_in, emb_in, config['relation_network_layers_config'], drop)
    for flag, net in (('freeze_featurizer', featurizer_network), ('freeze_attribute_network', attribute_network),
                      ('freeze_relation_network', relation_network), ('freeze_embedding_network', embedding_network)):
        if config.get(flag):
            net.requires_grad_(False)
    return {'featurizer_network': featurizer_network, 'attribute_network': attribute_network, 'relation_network': relation_network,
            'embedding_network': embedding_network, 'forward_attention_network': fwd_net, 'backward_attention_network': bwd_net,
            'attention_output_network': out_net}


def build_interpreter(config, neural_dict, ontology):           # gqa_interpreter_experiments.py:200-240
    featurizer = BatchGQABoxFeaturizer(featurizer_network=neural_dict['featurizer_network'])
    oracle = ClassifierOracle(ontology, neural_dict['attribute_network'], neural_dict['relation_network'], neural_dict['embedding_network'],
                              normalize=bool(config.get('normalize_oracle')), cached=True)
    if str(config.get('relation_tile_dtype', 'fp32')).lower() in ('bf16', 'bfloat16'):     # an extra key of this build (configs[4])
        oracle._tile_dtype = torch.bfloat16
    model = BatchGQAInterpreter(config['model_name'], oracle, ontology, featurizer, trainable_gate=config['trainable_gate'],
                                likelihood_threshold=config['likelihood_threshold'], hard_mode=config.get('hard_mode', False),
                                attention_transfer_state_dim=config['attention_transfer_state_dim'],
                                forward_attention_network=neural_dict['forward_attention_network'],
                                backward_attention_network=neural_dict['backward_attention_network'],
                                attention_output_network=neural_dict['attention_output_network'], cached=True)
    mlp_math = str(config.get('mlp_math', 'fp32')).lower()                                 # an extra key of this build (configs[3])
    if mlp_math in ('bf16', 'bfloat16'):
        model._mlp_math = 'bf16'
    elif mlp_math == 'bf16x3':
        # fp32 results from three exact bf16 pieces per operand (round 3's arithmetic): for object features whose scale is far from one - the
        # default forward products split their inputs into two UNSCALED fp16 pieces (error max(2^-22 |x|, 2^-25) per element)
        model._mlp_math = 'bf16x3'
    elif mlp_math not in ('fp32', 'float32', 'f32'):
        raise ValueError("mlp_math must be fp32, bf16x3 or bf16, got %r" % (config.get('mlp_math'),))
    return model


def build_model(config, ontology):                              # base_experiment.py:28-30
    return build_interpreter(config, build_neural_modules(config, ontology), ontology)
