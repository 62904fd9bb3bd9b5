"""GQA question JSON -> interpreter programs: the offline step in front of the program bytecode.

Restates the behaviour of the reference's `GQAPreprocessor` (src/gqa_preprocess.py:98-361) and `normalize`
(src/nsvqa/nn/parser/parse_utils.py:9-20); pinned on golden g15 (the reference's own outputs for questions authored in
tools/capture_goldens.py).  Host-side Python only - nothing here touches the GPU.

A GQA question carries its functional program as `semantic`: a list of {operation, argument, dependencies}.  The preprocessor
  1. maps every operation string through the operator map (`op_map.json`: "filter color" -> filter, "verify rel" -> verify_rel, ...;
     a question with an unmapped or null-mapped operation is dropped) and parses its argument string into the operator's arguments;
  2. merges `verify X` AND `verify Y` on the same object trace into one `verify_attrs [X, Y]`;
  3. batch format: cuts the program into select-rooted branches plus a terminal `last_op`, and rewrites the branch ends of and / or
     programs (exist dropped, verify_rel -> relate, verify_attrs -> filters);
  4. normalises the answer.

Word singularisation: the reference calls `pattern.text.en.singularize`, which this image does not have.  `normalize` takes the
singulariser as an argument; the default uses `pattern` when it is importable and otherwise `pattern_singularize`, a restatement of that
library's published algorithm (exception tables, then the first matching suffix rule).  The goldens were captured with the identity on a
vocabulary for which that is right, so everything except that one library call is pinned.
"""
import json
import os
import re

# words `normalize` leaves alone / maps itself (parse_utils.py:10-14)
PLURALE_TANTUM = frozenset((
    'this', 'yes', 'pants', 'shorts', 'glasses', 'scissors', 'panties', 'trousers', 'binoculars', 'pliers', 'tongs', 'tweezers',
    'forceps', 'goggles', 'jeans', 'tights', 'leggings', 'chaps', 'boxers', 'indoors', 'outdoors', 'bus', 'octapus', 'waitress',
    'pasta', 'pita', 'glass', 'asparagus', 'hummus', 'dress', 'cafeteria', 'grass', 'class'))
IRREGULAR = {'shelves': 'shelf', 'bookshelves': 'bookshelf', 'olives': 'olive', 'brownies': 'brownie', 'cookies': 'cookie'}


# ---- the singulariser --------------------------------------------------------------------------------------------------------------
# The reference calls `pattern.text.en.singularize` (parse_utils.py:6, 20; pattern is a declared dependency, setup.py, unpinned; the 3.6
# series is current for Python 3).  The library is not in this image, so its published algorithm (pattern/text/en/inflect.py: Conway-style
# rule list behind four exception tables) is RESTATED here: exception tables first - uninflected and uncountable words are returned as
# they are, -ie words lose their s, irregular plurals are replaced by table - then the first matching suffix rule wins.  The quirks of the
# original are kept (the uninflected / uncountable tests are `table_word.endswith(word)`, so any suffix of a table word passes through
# unchanged).  UNPINNED: no output of the library itself could be captured here; golden g15 uses a vocabulary for which the identity is right.
_SINGULAR_RULES = [(re.compile(rule), repl) for rule, repl in (
    (r'(?i)(.)ae$', r'\1a'), (r'(?i)(.)itis$', r'\1itis'), (r'(?i)(.)eaux$', r'\1eau'), (r'(?i)(quiz)zes$', r'\1'),
    (r'(?i)(matr)ices$', r'\1ix'), (r'(?i)(ap|vert|ind)ices$', r'\1ex'), (r'(?i)^(ox)en', r'\1'), (r'(?i)(alias|status)es$', r'\1'),
    (r'(?i)([octop|vir])i$', r'\1us'), (r'(?i)(cris|ax|test)es$', r'\1is'), (r'(?i)(shoe)s$', r'\1'), (r'(?i)(o)es$', r'\1'),
    (r'(?i)(bus)es$', r'\1'), (r'(?i)([m|l])ice$', r'\1ouse'), (r'(?i)(x|ch|ss|sh)es$', r'\1'), (r'(?i)(m)ovies$', r'\1ovie'),
    (r'(?i)(.)ombies$', r'\1ombie'), (r'(?i)(s)eries$', r'\1eries'), (r'(?i)([^aeiouy]|qu)ies$', r'\1y'),
    # -f, -fe sometimes take -ves in the plural (lives, wolves)
    (r'([aeo]l)ves$', r'\1f'), (r'([^d]ea)ves$', r'\1f'), (r'arves$', 'arf'), (r'erves$', 'erve'), (r'([nlw]i)ves$', r'\1fe'),
    (r'(?i)([lr])ves$', r'\1f'), (r'([aeo])ves$', r'\1ve'), (r'(?i)(sive)s$', r'\1'), (r'(?i)(tive)s$', r'\1'), (r'(?i)(hive)s$', r'\1'),
    (r'(?i)([^f])ves$', r'\1fe'),
    # -ses
    (r'(?i)(^analy)ses$', r'\1sis'), (r'(?i)((a)naly|(b)a|(d)iagno|(p)arenthe|(p)rogno|(s)ynop|(t)he)ses$', r'\1\2sis'),
    (r'(?i)(.)opses$', r'\1opsis'), (r'(?i)(.)yses$', r'\1ysis'), (r'(?i)(h|d|r|o|n|b|cl|p)oses$', r'\1ose'),
    (r'(?i)(fruct|gluc|galact|lact|ket|malt|rib|sacchar|cellul)ose$', r'\1ose'), (r'(?i)(.)oses$', r'\1osis'),
    # -a
    (r'(?i)([ti])a$', r'\1um'), (r'(?i)(n)ews$', r'\1ews'), (r'(?i)s$', ''))]
_SINGULAR_UNINFLECTED = frozenset((
    "bison", "debris", "headquarters", "pincers", "trout", "bream", "diabetes", "herpes", "pliers", "tuna", "breeches", "djinn", "high-jinks",
    "proceedings", "whiting", "britches", "eland", "homework", "rabies", "wildebeest", "carp", "elk", "innings", "salmon", "chassis", "flounder",
    "jackanapes", "scissors", "christmas", "gallows", "mackerel", "series", "clippers", "georgia", "measles", "shears", "cod", "graffiti", "mews",
    "species", "contretemps", "mumps", "swine", "corps", "news", "swiss"))
_SINGULAR_UNCOUNTABLE = frozenset((
    "advice", "equipment", "happiness", "luggage", "news", "software", "bread", "fruit", "information", "mathematics", "oil", "understanding",
    "butter", "furniture", "ketchup", "mayonnaise", "research", "water", "cheese", "garbage", "knowledge", "meat", "rice", "electricity", "gravel",
    "love", "mustard", "sand"))
_SINGULAR_IE = frozenset((
    "alergie", "cutie", "hoagie", "newbie", "softie", "veggie", "auntie", "doggie", "hottie", "nightie", "sortie", "weenie", "beanie", "eyrie",
    "indie", "oldie", "stoolie", "yuppie", "birdie", "foodie", "junkie", "pie", "sweetie", "zombie", "bogie", "genie", "laddie", "pixie", "techie",
    "bombie", "groupie", "laramie", "quickie", "tie", "collie", "hankie", "lingerie", "reverie", "toughie", "cookie", "hippie", "meanie", "rookie",
    "valkyrie"))
_SINGULAR_IRREGULAR = {
    "atlantes": "atlas", "atlases": "atlas", "axes": "axe", "beeves": "beef", "brethren": "brother", "children": "child", "corpora": "corpus",
    "corpuses": "corpus", "ephemerides": "ephemeris", "feet": "foot", "ganglia": "ganglion", "geese": "goose", "genera": "genus", "genii": "genie",
    "graffiti": "graffito", "helves": "helve", "kine": "cow", "leaves": "leaf", "loaves": "loaf", "men": "man", "mongooses": "mongoose",
    "monies": "money", "moves": "move", "mythoi": "mythos", "numena": "numen", "occipita": "occiput", "octopodes": "octopus", "opera": "opus",
    "opuses": "opus", "our": "my", "oxen": "ox", "penes": "penis", "penises": "penis", "people": "person", "sexes": "sex",
    "soliloquies": "soliloquy", "teeth": "tooth", "testes": "testis", "trilbys": "trilby", "turves": "turf", "zoa": "zoon"}
_PLURAL_PREPOSITIONS = frozenset(("about", "before", "during", "of", "till", "above", "behind", "except", "off", "to", "across", "below", "for",
                                  "on", "under", "after", "beneath", "from", "onto", "until", "among", "beside", "in", "out", "unto", "around",
                                  "besides", "into", "over", "upon", "at", "between", "near", "since", "with", "athwart", "betwixt", "beyond",
                                  "but", "by"))


def pattern_singularize(word):
    """pattern.text.en.singularize (nouns), restated - see the note above."""
    if "-" in word:                                        # compound words: mothers-in-law
        parts = word.split("-")
        if len(parts) > 1 and parts[1] in _PLURAL_PREPOSITIONS:
            return pattern_singularize(parts[0]) + "-" + "-".join(parts[1:])
    if word.endswith("'"):                                 # dogs' -> dog's
        return pattern_singularize(word[:-1]) + "'s"
    w = word.lower()
    if any(x.endswith(w) for x in _SINGULAR_UNINFLECTED) or any(x.endswith(w) for x in _SINGULAR_UNCOUNTABLE):
        return word
    for x in _SINGULAR_IE:
        if w.endswith(x + "s"):
            # pattern 3.x returns the lower-cased word UNCHANGED here (`return w`: the loop variable of pattern 2.6, the singular, became the
            # word when the function was refactored).  Not confirmable in this image - the library is absent - but the reference's own
            # `irregulars` table (parse_utils.py:14) lists 'cookies' and 'brownies', two entries of this list, which it would not need if
            # the library singularised them; its setup.py pins pattern >= 3.6.  Vocabularies built by the reference carry the quirk.
            return w
    for x, singular in _SINGULAR_IRREGULAR.items():
        if w.endswith(x):
            return re.sub('(?i)' + x + '$', singular, word)
    for rule, repl in _SINGULAR_RULES:
        m = rule.search(word)
        if m:
            for k, g in enumerate(m.groups()):
                if g is None:
                    repl = repl.replace('\\' + str(k + 1), '')
            return rule.sub(repl, word)
    return word


def default_singularize():
    try:
        from pattern.text.en import singularize          # the reference's choice
        return singularize
    except Exception:
        return pattern_singularize


def normalize(string, singularize=None):
    """parse_utils.py:9-20: lower-case, strip; irregulars by table; leave plurale tantum and '...ss' alone; singularise the rest."""
    word = string.strip().lower()
    if word in IRREGULAR:
        return IRREGULAR[word]
    if word.split(' ')[-1] in PLURALE_TANTUM or word[-2:] == 'ss':
        return word
    return (singularize or default_singularize())(word)


_ID_SUFFIX = re.compile(r'\((\d|,|\s)+\)|\((-|\s)*\)')          # "(1234)", "(12,34)", "(-)" after an object name


class GQAPreprocessor(object):
    """gqa_preprocess.py:98-361.  `map_json_path`: the operator map; `is_batch_format`: branches / last_op (what the bytecode encoder
    reads) instead of flat operator / argument / dependency lists."""

    STARTER_OPS = ('select',)
    TRACE_CHANGER_OPS = ('relate',)
    LOGICAL_OPS = ('and', 'or')

    def __init__(self, map_json_path, is_batch_format, singularize=None):
        with open(map_json_path, 'r') as f:
            self._op_map = json.load(f)
        self._is_batch_format = is_batch_format
        self._singularize = singularize or default_singularize()

    def _norm(self, token):
        return normalize(token, self._singularize)

    # ---- one operation ----------------------------------------------------------------------------
    def _arguments(self, op, op_tokens, arg_tokens):
        """Arguments of operator `op` from the words of the operation string and the comma-separated argument (:292-361)."""
        n = self._norm
        if op in ('select', 'filter', 'query_attr', 'verify_attr', 'all_same', 'all_different'):
            return (n(arg_tokens[0]),)
        if op in ('relate', 'verify_rel'):                       # "object name,relation,s|o": (relation, is-subject flag, name)
            return (n(arg_tokens[1]), arg_tokens[2] == 's', n(arg_tokens[0]))
        if op == 'choose_attr':
            return ([n(t) for t in arg_tokens[0].split('|')],)
        if op == 'verify_attrs':
            return ([n(t) for t in arg_tokens],)
        if op == 'choose_rel':
            return ([n(r) for r in arg_tokens[1].split('|')], arg_tokens[2] == 's', n(arg_tokens[0]))
        if op in ('exist', 'and', 'or', 'end'):
            return ()
        if op in ('two_same', 'two_different'):                  # "same color" -> "color"
            return (' '.join(n(t) for t in op_tokens[1:]),)
        if op == 'compare':                                      # "choose less healthy" / "choose older" -> (adjective, is_less)
            if len(op_tokens) >= 3:
                if n(op_tokens[1]) == 'more':
                    return (n(op_tokens[2]), False)
                if n(op_tokens[1]) == 'less':
                    return (n(op_tokens[2]), True)
            token = n(op_tokens[1])
            if token.endswith('er'):                             # comparative -> positive: older -> old, healthier -> healthy
                token = token[:-2]
                if token.endswith('i'):
                    token = token[:-1] + 'y'
            return (token, False)
        raise AttributeError("no argument parser for operator '%s'" % op)

    def parse_operation(self, operator, argument):
        op = self._op_map.get(operator)
        if op is None:
            return None, None
        arg = _ID_SUFFIX.sub('', argument).strip()
        return op, self._arguments(op, operator.split(' '), arg.split(','))

    def parse_program(self, program):
        parsed = [self.parse_operation(p['operation'], p['argument']) for p in program]
        # the dependency lists are the question's own objects: the merge below edits them in place, as the reference does
        return [o for o, _ in parsed], [a for _, a in parsed], [p['dependencies'] for p in program]

    # ---- program rewrites ---------------------------------------------------------------------------
    def _compute_op_trace(self, operators, dependencies):
        """Which object trace every operator works on: select and relate start a new one, everything else inherits (:215-226)."""
        trace, current = [], -1
        for op, dep in zip(operators, dependencies):
            if op in self.STARTER_OPS + self.TRACE_CHANGER_OPS:
                current += 1
                trace.append(current)
            else:
                trace.append(trace[dep[0]])
        return trace, current

    def _combine_verify(self, operators, arguments, dependencies, trace):
        """verify_attrs AND verify_attrs on one trace -> a single verify_attrs with both attributes (:228-249)."""
        last = dependencies[-1]
        if operators[-1] == 'and' and all(operators[i] == 'verify_attrs' for i in last) and trace[last[0]] == trace[last[1]]:
            first, second = min(last), max(last)
            for dep in dependencies:
                for j, d in enumerate(dep):
                    if d > first:
                        dep[j] = d - 1
            arguments[second] = [[arguments[first][0][0], arguments[second][0][0]]]
            for seq in (operators, arguments, dependencies, trace):
                del seq[first]
            return operators[:-1], arguments[:-1], dependencies[:-1], trace[:-1]
        return operators, arguments, dependencies, trace

    def _de_branch_program(self, operators, arguments, dependencies):
        """Branches rooted at each select, plus the terminal operator (:251-274)."""
        branch_of, branches = [], []
        for i in range(len(operators) - 1):
            if operators[i] in self.STARTER_OPS:
                branches.append([])
                branch_of.append(len(branches) - 1)
            elif dependencies[i] is not None and len(dependencies) > 0:
                branch_of.append(branch_of[dependencies[i][0]])
            elif i > 0:
                branch_of.append(branch_of[i - 1])
            else:
                raise ValueError('Operator not recognized.')
        for i in range(len(operators) - 1):
            branches[branch_of[i]].append({'operator': operators[i], 'arguments': arguments[i]})
        return {'branches': branches, 'last_op': {'operator': operators[-1], 'arguments': arguments[-1]}}

    def _fix_logical_branches(self, program):
        """Branches feeding and / or end in an attention, not a verdict (:197-213)."""
        if program['last_op']['operator'] in self.LOGICAL_OPS:
            for branch in program['branches']:
                tail = branch[-1]
                if tail['operator'] == 'exist':
                    del branch[-1]
                elif tail['operator'] == 'verify_rel':
                    tail['operator'] = 'relate'
                elif tail['operator'] == 'verify_attrs':
                    attrs = tail['arguments'][0]
                    tail['operator'], tail['arguments'] = 'filter', [attrs[0]]
                    branch.extend({'operator': 'filter', 'arguments': [a]} for a in attrs[1:])
        return program

    # ---- questions and files ----------------------------------------------------------------------------
    def parse_question(self, question, discard_global):
        """Adds the program (and the normalised answer) to `question`; None when the question is dropped (:167-190)."""
        head = question['semantic'][0]
        if discard_global and head['operation'] == 'select' and head['argument'] == 'scene':
            return None
        ops, args, deps = self.parse_program(question['semantic'])
        if None in ops or None in args:
            return None
        trace, _ = self._compute_op_trace(ops, deps)
        ops, args, deps, trace = self._combine_verify(ops, args, deps, trace)
        if self._is_batch_format:
            question['program'] = self._fix_logical_branches(self._de_branch_program(ops, args, deps))
        else:
            question['operators'], question['arguments'], question['dependencies'] = ops, args, deps
        question['answer'] = self._norm(question['answer'])
        return question

    @staticmethod
    def _dump_per_line(output, out_file):
        with open(out_file, 'a') as f:
            for value in output.values():
                f.write(json.dumps(value) + '\n')

    def preprocess(self, in_file, out_file, segregate, length_segregation, discard_global=False):
        """One JSON object per line, optionally one file per terminal operator (and per first-branch length) (:113-165)."""
        if os.path.isdir(in_file):
            file_names = [os.path.join(in_file, f) for f in os.listdir(in_file)
                          if os.path.isfile(os.path.join(in_file, f)) and (f.endswith('.json') or f.endswith('.txt'))]
        else:
            file_names = [in_file]
        stem, ext = os.path.splitext(out_file)
        for name in file_names:
            with open(name, 'r') as f:
                data = json.load(f)
            output = {}
            for key, value in data.items():
                if not isinstance(value, dict):
                    continue
                q = self.parse_question(value, discard_global)
                if q is None:
                    continue
                q['question_id'] = key
                if segregate:
                    op = q['program']['last_op']['operator'] if self._is_batch_format else q['operators'][-1]
                    if length_segregation:
                        op = op + '_' + str(len(q['program']['branches'][0]))
                    output.setdefault(op, {})[key] = q
                else:
                    output[key] = q
            if segregate:
                for op, group in output.items():
                    self._dump_per_line(group, stem + '_' + op + ext)
            else:
                self._dump_per_line(output, out_file)
