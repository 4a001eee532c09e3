"""Host-side mirror of the MATLAB class: quadrature, state tables, flags, code generation, error behaviour."""
import re

import numpy as np
import pytest

from egdst_amd import EgdstError, codegen, egdstmodel, examples, quadpoints
from egdst_amd.quadrature import quadrature_array


def test_quadpoints_are_gauss_legendre_on_unit_interval():
    # egdstmodel.m:1504-1529 -> Gauss-Legendre on [0,1] (not Gauss-Hermite)
    for n in (1, 2, 5, 10, 15, 21):
        x, w = quadpoints(n, 0, 1)
        xr, wr = np.polynomial.legendre.leggauss(n)
        assert np.allclose(np.sort(x), 0.5 * (xr + 1), atol=1e-15)
        assert np.allclose(w[np.argsort(x)], 0.5 * wr, atol=1e-15)
        assert abs(w.sum() - 1) < 1e-14
    q = quadrature_array(10)
    assert q.shape == (20,) and abs(q[:10].sum() - 1) < 1e-14 and np.all(np.diff(q[10:]) > 0)


def test_state_tables_first_variable_slowest():
    m = examples.retirement8(T=5, ngridm=10)
    assert (m.nst, m.nd, m.nnst, m.nnd) == (8, 2, 3, 1)
    assert m.stm == [2, 2, 2, 4, 2, 1]          # sizes + stepmult (egdstmodel.m:651,1432-1437)
    assert m.states[:, 0].tolist() == [0, 0, 0, 0, 1, 1, 1, 1]
    assert m.states[:, 2].tolist() == [0, 1] * 4
    assert m.indx([1, 0, 1]) == 6                # base-1 like the reference


def test_optim_flags_follow_compile_m():
    assert codegen.analyse_optim(examples.retirement2()) == dict(optim_MUnoD=True, optim_UnoD=False, optim_UasD=True,
                                                                 optim_TRPRnoSH=True)
    assert codegen.analyse_optim(examples.deaton2())['optim_UnoD'] is True
    assert codegen.analyse_optim(examples.occ3())['optim_UnoD'] is False


def test_numeric_trpr_and_discount_are_stringified_like_matlab():
    m = egdstmodel('x')
    m.s = ('st', [0, 'a', 1, 'b'])
    m.trpr = ('true', [[1 / 3, 2 / 3], [0.5, 0.5]])
    assert m.trpr[0].cases[0].prob[0] == ['0.3333333333', '0.6666666667']   # '%10.10f' (egdstmodel.m:1002)
    m.discount = 0.95
    assert m.discount == '%25.25f' % 0.95
    m.discount = 0
    assert m.discount == '0.0'


def test_setparam_getparam_and_ngridmax_rule():
    m = examples.retirement2()
    m.setparam('duw', 0.7)
    assert m.getparam('duw') == 0.7 and m.getparam(1) == 0.7
    m.setparam([0.1, 0.2, 0.3])
    assert m.getparam().tolist() == [0.1, 0.2, 0.3]
    with pytest.raises(EgdstError):
        m.setparam([1.0])
    with pytest.raises(EgdstError):
        m.setparam('nosuch', 1.0)
    m.ngridm = 900                                # 1.5*ngridm > ngridmax -> ngridmax = 2*ngridm (egdstmodel.m:532-538)
    assert m.ngridmax == 1800


def test_reference_error_behaviour():
    m = egdstmodel('e')
    with pytest.raises(EgdstError):
        m.solve()                                 # "needs to be compiled first" (egdstmodel.m:1142-1144)
    with pytest.raises(EgdstError):
        m.sim()
    m.s = ('s', [0, 'x'])
    m.d = ('d', [0, 'x'])
    with pytest.raises(codegen.CodegenError):     # Missing .u.utility (compile.m:271-273)
        codegen.generate_modelspec(m)
    m2 = examples.retirement2()
    with pytest.raises(EgdstError):
        m2.param = ('duw', 'duplicate ref', 1.0)  # checkrefs (egdstmodel.m:1463-1494)


def test_codegen_token_rewrite_and_banned_words():
    m = examples.occ3()
    txt = codegen.generate_modelspec(m)
    assert 'ms_coef_disutility[1][(int)ms_decisions[curr->id+0*MS_ND]+1]' in txt
    assert 'MS_POW(consumption,1-E->par[0])' in txt and 'MS_MAX(E->par[2],next->shock*0.5)' in txt
    assert re.search(r'0\.150000000000000', txt)  # coefficient arrays are printed %18.15f (compile.m:208)
    bad = examples.retirement2()
    bad.discount = '1/(1+interest)+0*id'
    with pytest.raises(codegen.CodegenError):     # ProhibitString: `id` not allowed in discount (compile.m:261)
        codegen.generate_modelspec(bad)
    bad2 = examples.retirement2()
    bad2.u = ('utility', 'log(consumption)+savings')
    with pytest.raises(codegen.CodegenError):
        codegen.generate_modelspec(bad2)


def test_nonseparable_utility_is_rejected():
    m = examples.retirement2()
    m.u = ('utility', 'log(consumption)*(1+duw*(id==0))')
    with pytest.raises(codegen.CodegenError):
        codegen.analyse_optim(m)


def test_call_needs_a_solved_model_and_known_names():
    """egdstmodel.m:1181-1207: call() refuses an unsolved model and unknown function names; the aliases map to the
    switch numbers of egdst_call.c."""
    from egdst_amd import examples, EgdstError
    from egdst_amd.model import egdstmodel
    m = examples.retirement2()
    with pytest.raises(EgdstError):
        m.call('utility', [[1, 1, 1, 1.0]])
    assert egdstmodel.CALL_NAMES['u'] == egdstmodel.CALL_NAMES['utility'] == 1
    assert egdstmodel.CALL_NAMES['mu'] == 2 and egdstmodel.CALL_NAMES['df'] == 3
    assert egdstmodel.CALL_NAMES['b'] == 4 and egdstmodel.CALL_NAMES['mb'] == 5 and egdstmodel.CALL_NAMES['vf'] == 6


def test_continuous_state_dsl_and_generated_plugin():
    """egdstmodel.m:628-646,997-1003 / compile.m:26-34,527-575: a continuous state is a linspace grid whose points are its
    values; its trpr entries are motion-rule strings; the plugin reads states by value when the simulator asks for it."""
    from egdst_amd import codegen, examples
    m = examples.retirement_hc()
    assert m.nst == 5 and m.nnst == 1 and m.s[0].type == 'continuous' and m.s[0].gridpoints == 5
    assert np.allclose(m.states[:, 0], np.linspace(0, 2, 5)) and m.stm == [5, 1]
    assert isinstance(m.trpr[0].cases[0].prob, str) and len(m.trpr[0].cases) == 2
    text = codegen.generate_modelspec(m)
    assert '#define MS_NCONT 1' in text and 'ms_trpr_cont' in text and 'ms_bxsearch(nval, ms_stgrid1, 5)' in text
    assert 'curr->byval>0?curr->st[0]:ms_states[curr->ist+0*MS_NST]' in text
    assert '#define MS_NCONT 0' in codegen.generate_modelspec(examples.retirement2())
    with pytest.raises(Exception):
        m.d = ('a continuous decision', [0, 1], 4)
