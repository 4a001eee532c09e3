"""An INDEPENDENT evaluation of a model's executable strings (test infrastructure).

The model plugin of the reference is C code generated from the user's strings by compile.m; here egdst_amd/codegen.py does
that, and the SAME generated modelspec.h feeds both the device library and the CPU oracle -- a rewriting bug would be common
to both and invisible to every parity test.  This module shares nothing with codegen.py: it parses the user's C expression
strings itself (a small Pratt parser) and evaluates them with the meanings the reference gives the DSL's identifiers
(StdConvertN, @egdstmodel/compile.m:12-64; defaults egdstmodel.m:392-393):

    it            period index, base 0                  age        it + t0
    id, ist       decision / state index, base 0        ist1       next period's state index
    dcK           value of decision variable K of decision id   (decisions[id + (K-1)*nd], compile.m:26-34)
    stK / stKn    value of state variable K of state ist / ist1
    cash          curr.cash        savings, shock   next.savings, next.shock
    mu, sigma     the shock parameters' own strings (mu_param / sigma_param of curr, next)
    discount, survival   their strings       <param ref>  the parameter value      <coef ref>[i][j]  base-1 tables
    <eq ref>      the equation's string      min, max     MIN / MAX macros           true, false  1, 0
    consumption / mutility / x   the argument of utility (marginal) / marginal inverse / the transformation

C semantics that matter: comparisons and logical operators yield int 0/1, `int / int` truncates, `(int)x` truncates,
`a ? b : c` evaluates one branch.  libm functions are Python's math module (the platform libm, as in the reference's MEX).
"""
import math
import re

_TOKEN = re.compile(r'\s*(?:(?P<num>(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)|(?P<id>[A-Za-z_][A-Za-z_0-9]*)|'
                    r'(?P<op>&&|\|\||==|!=|<=|>=|[-+*/%<>!?:(),\[\]]))')

_BINARY = {'||': 1, '&&': 2, '==': 3, '!=': 3, '<': 4, '<=': 4, '>': 4, '>=': 4, '+': 5, '-': 5, '*': 6, '/': 6, '%': 6}


def tokenize(text):
    pos, out = 0, []
    text = text.strip().rstrip(';')
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m or m.end() == pos:
            raise ValueError('cannot tokenize %r at %d' % (text, pos))
        pos = m.end()
        if m.group('num') is not None:
            t = m.group('num')
            out.append(('num', float(t) if re.search(r'[.eE]', t) else int(t)))
        elif m.group('id') is not None:
            out.append(('id', m.group('id')))
        else:
            out.append(('op', m.group('op')))
    return out


class _Parser:
    """expression -> nested tuples"""

    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else ('end', None)

    def take(self, kind=None, val=None):
        tok = self.peek()
        if (kind and tok[0] != kind) or (val is not None and tok[1] != val):
            raise ValueError('expected %s %s, got %s' % (kind, val, tok))
        self.i += 1
        return tok

    def ternary(self):
        c = self.binary(1)
        if self.peek() == ('op', '?'):
            self.take()
            a = self.ternary()
            self.take('op', ':')
            b = self.ternary()
            return ('?', c, a, b)
        return c

    def binary(self, minprec):
        left = self.unary()
        while True:
            tok = self.peek()
            if tok[0] != 'op' or tok[1] not in _BINARY or _BINARY[tok[1]] < minprec:
                return left
            self.take()
            right = self.binary(_BINARY[tok[1]] + 1)   # left-associative
            left = ('bin', tok[1], left, right)

    def unary(self):
        tok = self.peek()
        if tok == ('op', '-'):
            self.take()
            return ('neg', self.unary())
        if tok == ('op', '+'):
            self.take()
            return self.unary()
        if tok == ('op', '!'):
            self.take()
            return ('not', self.unary())
        if tok == ('op', '(') and self.i + 2 < len(self.t) and self.t[self.i + 1] in (('id', 'int'), ('id', 'double')) \
                and self.t[self.i + 2] == ('op', ')'):
            kind = self.t[self.i + 1][1]
            self.i += 3
            return ('cast', kind, self.unary())
        return self.postfix()

    def postfix(self):
        tok = self.take()
        if tok[0] == 'num':
            node = ('num', tok[1])
        elif tok[0] == 'id':
            node = ('id', tok[1])
        elif tok == ('op', '('):
            node = self.ternary()
            self.take('op', ')')
        else:
            raise ValueError('unexpected %s' % (tok,))
        while True:
            nxt = self.peek()
            if nxt == ('op', '(') and node[0] == 'id':
                self.take()
                args = []
                if self.peek() != ('op', ')'):
                    args.append(self.ternary())
                    while self.peek() == ('op', ','):
                        self.take()
                        args.append(self.ternary())
                self.take('op', ')')
                node = ('call', node[1], args)
            elif nxt == ('op', '['):
                self.take()
                idx = self.ternary()
                self.take('op', ']')
                node = ('index', node, idx)
            else:
                return node


def parse(text):
    p = _Parser(tokenize(text))
    tree = p.ternary()
    if p.peek()[0] != 'end':
        raise ValueError('trailing tokens in %r' % text)
    return tree


_LIBM = {'log': math.log, 'exp': math.exp, 'pow': math.pow, 'sqrt': math.sqrt, 'fabs': math.fabs, 'floor': math.floor,
         'ceil': math.ceil, 'log1p': math.log1p, 'expm1': math.expm1, 'tanh': math.tanh, 'sin': math.sin, 'cos': math.cos,
         'atan': math.atan, 'erf': math.erf, 'erfc': math.erfc, 'fmod': math.fmod}


def _c_log(x):
    if x == 0:
        return -math.inf
    if x < 0 or x != x:
        return math.nan
    return math.log(x)


def _c_pow(x, y):
    try:
        return math.pow(x, y)
    except (OverflowError, ValueError, ZeroDivisionError):
        if x == 0 and y < 0:
            return math.inf
        return math.nan


def _c_div(a, b):
    if isinstance(a, int) and isinstance(b, int):
        q = abs(a) // abs(b)
        return q if (a >= 0) == (b >= 0) else -q
    if b == 0:
        if a == 0 or a != a:
            return math.nan
        return math.copysign(math.inf, a) * math.copysign(1.0, b)
    return a / b


class Coef:
    """a coefficient table with the reference's base-1 indexing (compile.m:199-219 pads row and column 0)"""

    def __init__(self, array):
        self.a = array

    def __getitem__(self, i):
        if i == 0:
            return _Row(None)
        return _Row(self.a[i - 1])


class _Row:
    def __init__(self, row):
        self.r = row

    def __getitem__(self, j):
        return 0.0 if (self.r is None or j == 0) else float(self.r[j - 1])


class PluginEval:
    """Evaluates the strings of one egdstmodel for given (curr, next) period variables and a parameter vector."""

    def __init__(self, model, params=None):
        self.m = model
        self.par = {p.ref: float(v) for p, v in zip(model.param, model.param_vector() if params is None else params)}
        self.coef = {c.ref: Coef(c.array) for c in model.coef}
        self.eq = {e.ref: e for e in model.eq}
        self._trees = {}

    def tree(self, text):
        if text not in self._trees:
            self._trees[text] = parse(text)
        return self._trees[text]

    # -- the DSL's identifiers --------------------------------------------------------------------------------------
    def ident(self, name, cur, nxt, extra):
        m = self.m
        if name in extra:
            return extra[name]
        if name == 'it':
            return cur['it']
        if name == 'age':
            return cur['it'] + int(m.t0)
        if name == 'id':
            return cur['id']
        if name == 'ist':
            return cur['ist']
        if name == 'ist1':
            return nxt['ist']
        mm = re.fullmatch(r'dc(\d+)', name)
        if mm and 1 <= int(mm.group(1)) <= m.nnd:
            return float(m.decisions[cur['id']][int(mm.group(1)) - 1])
        mm = re.fullmatch(r'st(\d+)(n?)', name)
        if mm and 1 <= int(mm.group(1)) <= m.nnst:
            who = nxt if mm.group(2) else cur
            return float(m.states[who['ist']][int(mm.group(1)) - 1])
        if name == 'cash':
            return cur['cash']
        if name == 'savings':
            return nxt['savings']
        if name == 'shock':
            return nxt['shock']
        if name == 'mu':
            return self.eval(m.shock['mu'], cur, nxt)
        if name == 'sigma':
            return self.eval(m.shock['sigma'], cur, nxt)
        if name == 'discount':
            return self.eval(m.discount, cur, nxt)
        if name == 'survival':
            return self.eval(m.survival, cur, nxt)
        if name in self.par:
            return self.par[name]
        if name in self.coef:
            return self.coef[name]
        if name in self.eq:
            return self.eval(self.eq[name].expression, cur, nxt)
        if name == 'true':
            return 1
        if name == 'false':
            return 0
        if name in ('t0', 'T', 'ngridm', 'ngridmax', 'nthrhmax', 'ny'):
            return int(getattr(m, name))
        if name in ('mmax', 'a0'):
            return float(getattr(m, name))
        if name in ('nd', 'nnd', 'nst', 'nnst'):
            return int(getattr(m, name))
        if name == 'INFINITY':
            return math.inf
        raise KeyError('identifier %r has no meaning in the model DSL' % name)

    def eval(self, text, cur, nxt=None, **extra):
        if not isinstance(text, str):
            raise TypeError('multi-statement bodies are not used by the models under test')
        return self._ev(self.tree(text), cur, nxt, extra)

    def _ev(self, n, cur, nxt, extra):
        k = n[0]
        if k == 'num':
            return n[1]
        if k == 'id':
            return self.ident(n[1], cur, nxt, extra)
        if k == 'neg':
            return -self._ev(n[1], cur, nxt, extra)
        if k == 'not':
            return 0 if self._ev(n[1], cur, nxt, extra) else 1
        if k == 'cast':
            v = self._ev(n[2], cur, nxt, extra)
            return int(v) if n[1] == 'int' else float(v)      # int(): truncation towards zero, as the C cast
        if k == '?':
            return self._ev(n[2] if self._ev(n[1], cur, nxt, extra) else n[3], cur, nxt, extra)
        if k == 'index':
            return self._ev(n[1], cur, nxt, extra)[self._ev(n[2], cur, nxt, extra)]
        if k == 'call':
            args = [self._ev(a, cur, nxt, extra) for a in n[2]]
            f = n[1]
            if f == 'max':
                return args[0] if args[0] > args[1] else args[1]      # MAX(X,Y) (((X)>(Y))?(X):(Y))
            if f == 'min':
                return args[0] if args[0] < args[1] else args[1]
            if f == 'log':
                return _c_log(float(args[0]))
            if f == 'pow':
                return _c_pow(float(args[0]), float(args[1]))
            return _LIBM[f](*[float(a) for a in args])
        if k == 'bin':
            op = n[1]
            if op == '&&':
                return 1 if (self._ev(n[2], cur, nxt, extra) and self._ev(n[3], cur, nxt, extra)) else 0
            if op == '||':
                return 1 if (self._ev(n[2], cur, nxt, extra) or self._ev(n[3], cur, nxt, extra)) else 0
            a, b = self._ev(n[2], cur, nxt, extra), self._ev(n[3], cur, nxt, extra)
            if op == '+':
                return a + b
            if op == '-':
                return a - b
            if op == '*':
                return a * b
            if op == '/':
                return _c_div(a, b)
            if op == '%':
                return int(math.fmod(a, b))
            return {'==': a == b, '!=': a != b, '<': a < b, '<=': a <= b, '>': a > b, '>=': a >= b}[op] and 1 or 0
        raise ValueError(n)

    # -- the model functions of the plugin (compile.m:255-551) ----------------------------------------------------------
    def utility(self, cur, c):
        return float(self.eval(self.m.u['utility'], cur, None, consumption=c))

    def utility_marginal(self, cur, c):
        return float(self.eval(self.m.u['marginal'], cur, None, consumption=c))

    def utility_marginal_inverse(self, cur, mu):
        return float(self.eval(self.m.u['marginalinverse'], cur, None, mutility=mu))

    def discount(self, cur):
        return float(self.eval(self.m.discount, cur, None))

    def survival(self, cur):
        return float(self.eval(self.m.survival, cur, None))

    def cashinhand(self, cur, nxt):
        return float(self.eval(self.m.budget['cashinhand'], cur, nxt))

    def cashinhand_marginal(self, cur, nxt):
        return float(self.eval(self.m.budget['marginal'], cur, nxt))

    def mu(self, cur, nxt):
        return float(self.eval(self.m.shock['mu'], cur, nxt))

    def sigma(self, cur, nxt):
        return float(self.eval(self.m.shock['sigma'], cur, nxt))

    def equation(self, ref, cur, nxt=None):
        return float(self.eval(self.eq[ref].expression, cur, nxt))

    def feasible(self, cur):
        res = bool(self.m.feasible['defaultfeasible'])
        for r in self.m.feasible['rules']:
            if self.eval(r['condition'], cur, None):
                res = not self.m.feasible['defaultfeasible']
        return int(res)

    def inchoiceset(self, cur):
        res = bool(self.m.choiceset['defaultallow'])
        for r in self.m.choiceset['rules']:
            if self.eval(r['condition'], cur, None):
                res = not self.m.choiceset['defaultallow']
        return int(res)

    def trpr(self, cur, nxt):
        """transition probability of the discrete state variables (compile.m:476-551, all = 0)"""
        m = self.m
        sizes = [int(x) for x in m.stm[:m.nnst]]
        strides = [int(x) for x in m.stm[m.nnst:]]
        res = 1.0
        for tr in m.trpr:
            k = tr.varindex - 1
            if m.s[k].type == 'continuous':
                continue
            i, j = (cur['ist'] // strides[k]) % sizes[k], (nxt['ist'] // strides[k]) % sizes[k]
            for case in tr.cases:
                if self.eval(case.condition, cur, nxt):
                    p = case.prob[i][j]
                    res *= float(self.eval(p, cur, nxt)) if isinstance(p, str) else float(p)
                    break
            else:
                raise ValueError('trpr: the set of cases is not complete')
            if res == 0.0:
                return 0.0
        return res
