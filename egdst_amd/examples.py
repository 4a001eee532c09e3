"""The reference's example models restated through the Python mirror of the egdstmodel DSL.

Parameter values and functional forms are data taken from the reference scripts:
  deaton1/2      egdst_examples/model_deaton1.m:6-41, model_deaton2.m:6-41
  retirement1/2  egdst_examples/model_retirement1.m:6-42, model_retirement2.m:6-42
  occ3           egdst_examples/model_occ3.m:3-46
  model2         lecture_code/model2.m:4-78
  retirement8    SURVEY.md §8(d) config C5: model2's labour state x two binary exogenous Markov states
Keyword overrides (T, ngridm, ny, ...) produce the scaled BASELINE configs.
"""
from __future__ import annotations

from .model import egdstmodel


def _apply(m, over):
    for k, v in over.items():
        if k in ('t0', 'T', 'mmax', 'ngridmax', 'ngridm', 'nthrhmax', 'ny', 'a0'):
            setattr(m, k, v)
        else:
            m.setparam(k, v)
    return m


def _deaton(label, sigma, mu, a0_, mmax_, ny_, **over):
    m = egdstmodel(label)
    m.t0 = 1
    m.T = 25
    m.mmax = mmax_
    m.ngridmax = 1000
    m.ngridm = 100
    m.nthrhmax = 10
    m.ny = ny_
    m.s = ('Singleton state', [0, 'dummy state'])
    m.trpr = ('true', [[1]])
    m.feasible = ('defaultfeasible', True)
    m.d = ('Dummy decision', [0, 'dummy decision'])
    m.choiceset = ('defaultallow', True)
    m.u = ('utility', 'log(consumption)')
    m.u = ('marginal', '1/consumption')
    m.u = ('marginalinverse', '1/mutility')
    m.u = ('extrap', 'log(x)')
    m.budget = ('cashinhand', 'savings*(1+interest)+income_level')
    m.budget = ('marginal', '1+interest')
    m.discount = '1/(1+interest)'
    m.param = ('interest', 'return on savings', 0.01)
    m.eq = ('income_level', 'Realized income', 'income*shock', 'next')
    m.param = ('income', 'income (times multiplicator shock)', 1.25)
    m.a0 = a0_
    m.shock = 'lognormal'
    m.shock = ('sigma', sigma)
    m.shock = ('mu', mu)
    return _apply(m, over)


def deaton1(**over):
    return _deaton('deaton1', '0', '0', 0, 50, 2, **over)


def deaton2(**over):
    return _deaton('deaton2', '0.75', '-0.5*sigma*sigma', -25, 100, 10, **over)


def deaton_sig(**over):
    """Deaton forms with sigma as a run-time parameter (draws over interest, income, sigma)."""
    m = _deaton('deatonsig', 'sig', '-0.5*sigma*sigma', 0, 50, 5)
    m.param = ('sig', 'sigma of the lognormal income shock', 0.75)
    return _apply(m, over)


def _retirement(label, sigma, mu, **over):
    m = egdstmodel(label)
    m.t0 = 1
    m.T = 25
    m.mmax = 10
    m.ngridmax = 1000
    m.ngridm = 100
    m.nthrhmax = 10
    m.ny = 10
    m.s = ('Singleton state', [0, 'dummy state'])
    m.trpr = ('true', [[1]])
    m.feasible = ('defaultfeasible', True)
    m.d = ('Labour supply', [0, 'retire', 1, 'work'])
    m.choiceset = ('defaultallow', True)
    m.u = ('utility', 'log(consumption)+duw*(id==0)')
    m.param = ('duw', 'disutility of work', 0.5)
    m.u = ('marginal', '1/consumption')
    m.u = ('marginalinverse', '1/mutility')
    m.u = ('extrap', 'log(x)')
    m.budget = ('cashinhand', 'savings+wage_income*(id!=0)')
    m.budget = ('marginal', '1+interest')
    m.discount = '1/(1+interest)'
    m.param = ('interest', 'return on savings', 0.045)
    m.eq = ('wage_income', 'Realized wage income', 'wage*shock', 'next')
    m.param = ('wage', 'wage (times multiplicator shock)', 1.05)
    m.a0 = -5
    m.shock = 'lognormal'
    m.shock = ('sigma', sigma)
    m.shock = ('mu', mu)
    return _apply(m, over)


def retirement1(**over):
    return _retirement('retire1', '0', '0', **over)


def retirement2(**over):
    return _retirement('retire2', '0.25', '-0.5*sigma*sigma', **over)


def retirement_sig(**over):
    """retirement2 forms with sigma as a run-time parameter (draws over duw, wage, sigma)."""
    m = _retirement('retiresig', 'sig', '-0.5*sigma*sigma')
    m.param = ('sig', 'sigma of the lognormal wage shock', 0.25)
    return _apply(m, over)


def occ3(**over):
    m = egdstmodel('Occupational choice model ')
    m.t0 = 0
    m.T = 40
    m.s = ('Dummy state', [0, 'dummy'])
    m.trpr = ('true', [[1]])
    m.feasible = ('defaultfeasible', True)
    m.d = ('Occupational choice', [0, 'public sector (lower pay, secure)',
                                   1, 'private sector (hight pay, less secure)',
                                   2, 'entrepreneurship'])
    m.choiceset = ('defaultallow', True)
    m.u = ('utility', '(pow(consumption,1-crra)-1)/(1-crra) - coefleisure*disutility[1][(int)dc1+1]')
    m.coef = ('disutility', 'Disutility of work', [[0.0, 1.0, 0.75]])
    m.param = ('crra', 'CRRA coefficient in utility', 1.2)
    m.param = ('coefleisure', 'Weight with leisure in utility', 0.2)
    m.u = ('marginal', 'pow(consumption,-crra)')
    m.u = ('marginalinverse', 'pow(mutility,-1/crra)')
    m.u = ('extrap', 'pow(x,1-crra)')
    m.discount = '0.93'
    m.shock = 'lognormal'
    m.shock = ('sigma', 'sigs[1][(int)dc1+1]')
    m.coef = ('sigs', 'Sigmas for different occupations', [[0.15, 0.35, 0.75]])
    m.shock = ('mu', '-0.5*sigs[1][(int)dc1+1]*sigs[1][(int)dc1+1]')
    m.eq = ('wage1', 'Realized wage in the public sector', 'max(ssinc,shock*0.5)', 'next')
    m.eq = ('wage2', 'Realized wage in the private sector', 'max(ssinc,shock*0.5*wagegap)', 'next')
    m.param = ('ssinc', 'Guaranteed social security income', 0.01)
    m.param = ('wagegap', 'Wage gap between public and private sector', 1.35)
    m.eq = ('entrep', 'Entrepreneurial income (realized)', 'max(ssinc,log(savings+1)*entrkap*shock)', 'next')
    m.param = ('entrkap', 'Return on capital', 0.56)
    m.budget = ('cashinhand', 'savings*(1+interest)+(dc1==0)*wage1+(dc1==1)*wage2+(dc1==2)*entrep')
    m.budget = ('marginal', '1+interest+(dc1==2)*max(0,entrkap*shock/(savings+1))')
    m.param = ('interest', 'return on savings', 0.05)
    m.a0 = 0
    m.mmax = 5
    m.ngridm = 50
    m.ngridmax = 100
    m.nthrhmax = 100
    m.ny = 10
    return _apply(m, over)


def _model2_parts(m):
    m.u = ('utility', '(fabs(rho)<1e-10?log(consumption):(pow(consumption,rho)-1)/rho)  - (id?duw:0.0)')
    m.u = ('marginal', 'pow(consumption,rho-1)')
    m.u = ('marginalinverse', 'pow(mutility,1/(rho-1))')
    m.u = ('extrap', 'pow(x,rho)')
    m.budget = ('cashinhand', 'savings*(1+r)*shock  + (id?wage:0.0)')
    m.budget = ('marginal', '(1+r)*shock')
    m.discount = 'df'
    m.shock = 'lognormal'
    m.shock = ('sigma', 'sig')
    m.shock = ('mu', '-sigma*sigma/2')


def model2(T=3, ngridm=100, nquad=10, mmax=100.0, cc=0.0, df=1.0, rho=0.0, r=0.0, sigma=0.0, duw=1.0,
           wage=5.0):
    """lecture_code/model2.m: 2 labour states x 2 decisions, retirement absorbing."""
    m = egdstmodel('model2 for ZICE2014 lecture')
    m.t0 = 1
    m.T = T
    m.mmax = mmax
    m.ngridmax = 10 * ngridm
    m.ngridm = ngridm
    m.nthrhmax = ngridm
    m.ny = nquad
    m.a0 = 0
    m.s = ('Labour market state', [0, 'retired', 1, 'working'])
    m.d = ('Retirement decision', [0, 'Retirement', 1, 'Work'])
    m.feasible = ('defaultfeasible', True)
    m.trpr = ('dc1==0', [[1, 0], [1, 0]])
    m.trpr = ('dc1==1', [[0, 1], [0, 1]])
    m.choiceset = ('defaultallow', True)
    m.choiceset = ('ist==0 && id==1', 'Retirement is absorbing')
    m.param = ('rho', '1-crra parameter', rho)
    m.param = ('duw', 'scale parameter for disutility of work', duw)
    m.param = ('r', 'risk free return', r)
    m.param = ('df', 'discount factor', df)
    m.param = ('wage', 'Workers wage', wage)
    m.param = ('sig', 'sigma parameter in lognormal return', sigma)
    _model2_parts(m)
    m.a0 = cc
    return m


def retirement8(T=100, ngridm=32768, ny=15, mmax=50.0, duw=0.5, wage=1.05, sigma=0.25, interest=0.045):
    """nst=8 (labour state x health x wage regime), nd=2, retirement absorbing.

    Labour state follows the decision (lecture_code/model2.m:34-43); the two extra binary states are
    exogenous Markov chains with the row-stochastic matrix [[.9,.1],[.2,.8]].  Wage income is scaled by
    the wage regime, utility of retirement leisure by health; functional forms otherwise as
    model_retirement2.m with a0=0.
    """
    m = egdstmodel('retire8')
    m.t0 = 1
    m.T = T
    m.mmax = mmax
    m.ngridmax = 10 * ngridm
    m.ngridm = ngridm
    m.nthrhmax = ngridm
    m.ny = ny
    m.a0 = 0
    m.s = ('Labour market state', [0, 'retired', 1, 'working'])
    m.trpr = ('dc1==0', [[1, 0], [1, 0]])
    m.trpr = ('dc1==1', [[0, 1], [0, 1]])
    m.s = ('Health', [0, 'poor', 1, 'good'])
    m.trpr = ('true', [[0.9, 0.1], [0.2, 0.8]])
    m.s = ('Wage regime', [0, 'low', 1, 'high'])
    m.trpr = ('true', [[0.9, 0.1], [0.2, 0.8]])
    m.d = ('Retirement decision', [0, 'Retirement', 1, 'Work'])
    m.feasible = ('defaultfeasible', True)
    m.choiceset = ('defaultallow', True)
    m.choiceset = ('st1==0 && id==1', 'Retirement is absorbing')
    m.param = ('duw', 'disutility of work', duw)
    m.param = ('interest', 'return on savings', interest)
    m.param = ('wage', 'wage (times multiplicator shock)', wage)
    m.param = ('sig', 'sigma of the lognormal wage shock', sigma)
    m.u = ('utility', 'log(consumption)+duw*(1.0+0.2*st2)*(id==0)')
    m.u = ('marginal', '1/consumption')
    m.u = ('marginalinverse', '1/mutility')
    m.u = ('extrap', 'log(x)')
    m.eq = ('wage_income', 'Realized wage income', 'wage*(1.0+0.25*st3n)*shock', 'next')
    m.budget = ('cashinhand', 'savings*(1+interest)+wage_income*(id!=0)')
    m.budget = ('marginal', '1+interest')
    m.discount = '1/(1+interest)'
    m.shock = 'lognormal'
    m.shock = ('sigma', 'sig')
    m.shock = ('mu', '-0.5*sigma*sigma')
    return m



def cake_normal(**over):
    """egdst_examples/model_cake1.m:6-40 forms (log utility, one state, one decision, discount 1) with the shock switched
    on: an ADDITIVE NORMAL income (shock type 'normal': rescale() = mu + x*sigma, egdst_lib.c:84-100) -- the DISTRIB=2
    branch of the solver and the simulator, which none of the shipped scripts reaches with sigma > 0."""
    m = egdstmodel('cakenormal')
    m.t0 = 1
    m.T = 25
    m.mmax = 10
    m.ngridmax = 1000
    m.ngridm = 100
    m.nthrhmax = 10
    m.ny = 6
    m.s = ('Singleton state', [0, 'dummy state'])
    m.trpr = ('true', [[1]])
    m.feasible = ('defaultfeasible', True)
    m.d = ('Dummy decision', [0, 'dummy decision'])
    m.choiceset = ('defaultallow', True)
    m.u = ('utility', 'log(consumption)')
    m.u = ('marginal', '1/consumption')
    m.u = ('marginalinverse', '1/mutility')
    m.u = ('extrap', 'log(x)')
    m.budget = ('cashinhand', 'savings+shock')
    m.budget = ('marginal', '1')
    m.discount = '0.97'
    m.a0 = 0
    m.shock = 'normal'
    m.shock = ('sigma', '0.15')
    m.shock = ('mu', '0.6')
    return _apply(m, over)


def retirement_mortal(**over):
    """model_retirement2.m forms with a survival probability below one that falls with age (egdstmodel.m:95-104): the
    simulator's death draw (egdst_simulator.c:265) -- agents leave the panel, the remaining periods stay NaN."""
    m = _retirement('retiremortal', '0.25', '-0.5*sigma*sigma')
    m.survival = '1.0-0.004*it'
    return _apply(m, over)


def retirement_hc(**over):
    """model_retirement2.m forms plus a CONTINUOUS state -- human capital on a 5-point grid over [0, 2]
    that grows while working and depreciates otherwise (deterministic motion rules, egdstmodel.m:166-169) and scales the
    wage.  No shipped script has a continuous state; this one exercises trpr(..., all=1) in the solver
    (compile.m:527-538) and the 2^k corner blend of the simulator (egdst_simulator.c:313-372)."""
    m = egdstmodel('retirehc')
    m.t0 = 1
    m.T = 20
    m.mmax = 10
    m.ngridmax = 1000
    m.ngridm = 80
    m.nthrhmax = 100
    m.ny = 5
    m.s = ('Human capital', [0.0, 2.0], 5)
    m.trpr = ('dc1==1', 'min(st1*0.9+0.3,2.0)')
    m.trpr = ('dc1==0', 'st1*0.9')
    m.feasible = ('defaultfeasible', True)
    m.d = ('Labour supply', [0, 'retire', 1, 'work'])
    m.choiceset = ('defaultallow', True)
    m.u = ('utility', 'log(consumption)+duw*(id==0)')
    m.param = ('duw', 'disutility of work', 0.5)
    m.u = ('marginal', '1/consumption')
    m.u = ('marginalinverse', '1/mutility')
    m.u = ('extrap', 'log(x)')
    m.budget = ('cashinhand', 'savings*(1+interest)+wage_income*(id!=0)')
    m.budget = ('marginal', '1+interest')
    m.discount = '1/(1+interest)'
    m.param = ('interest', 'return on savings', 0.045)
    m.eq = ('wage_income', 'Realized wage income', 'wage*(0.6+0.4*st1n)*shock', 'next')
    m.param = ('wage', 'wage (times multiplicator shock)', 1.05)
    m.a0 = 0
    m.shock = 'lognormal'
    m.shock = ('sigma', '0.25')
    m.shock = ('mu', '-0.5*sigma*sigma')
    return _apply(m, over)


REGISTRY = {'deaton1': deaton1, 'deaton2': deaton2, 'deaton_sig': deaton_sig, 'retirement1': retirement1,
            'retirement2': retirement2, 'retirement_sig': retirement_sig, 'occ3': occ3, 'model2': model2,
            'retirement8': retirement8, 'cake_normal': cake_normal, 'retirement_mortal': retirement_mortal, 'retirement_hc': retirement_hc}
