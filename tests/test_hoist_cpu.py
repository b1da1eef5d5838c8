"""Uniform hoisting (pde_amd/stencil/expr.py: hoist_uniforms) must not change what the generated code computes."""
import random

import pytest

from pde_amd.stencil import expr as X
from pde_amd.stencil.specs import SPECS
from pde_amd.stencil.codegen import split_adjoint


def _value(e, env, table=None):
    """scalar evaluation with random but fixed values for every leaf; predicates are random but fixed booleans"""
    if isinstance(e, X.Num):
        return e.v
    if isinstance(e, X.Uni):
        return _value(table.exprs[e.i], env, table)
    if isinstance(e, (X.Par, X.Ref, X.Aux, X.GRef)):
        return env.setdefault(e.key(), random.uniform(0.5, 1.5) * random.choice((-1, 1)))
    if isinstance(e, X.Add):
        return sum(_value(t, env, table) for t in e.terms)
    if isinstance(e, X.Mul):
        out = 1.0
        for t in e.factors:
            out *= _value(t, env, table)
        return out
    if isinstance(e, X.Recip):
        return 1.0 / _value(e.e, env, table)
    if isinstance(e, X.Sel):
        return _value(e.a, env, table) if _value(e.c, env, table) >= 0 else _value(e.b, env, table)
    if isinstance(e, X.PSel):
        return _value(e.a, env, table) if env.setdefault(e.p.key(), random.random() < 0.5) else _value(e.b, env, table)
    raise TypeError(type(e))


@pytest.mark.parametrize("name", sorted(SPECS))
def test_hoisted_expressions_keep_their_value(name):
    s = SPECS[name]
    random.seed(hash(name) & 0xffff)
    tab = X.UniformTable()
    exprs = list(s.residuals) + list(s.adjoint)
    for e in s.adjoint:
        exprs += list(split_adjoint(e))
    hoisted = [X.hoist_uniforms(e, tab) for e in exprs]
    for trial in range(3):
        env = {}
        for e, h in zip(exprs, hoisted):
            a, b = _value(e, env), _value(h, env, tab)
            assert abs(a - b) <= 1e-9 * max(1.0, abs(a)), (name, a, b)
    # every table entry is uniform (literals, constants, per-sample parameters only) and not a bare leaf
    for u in tab.exprs:
        assert X.is_uniform(u) and not isinstance(u, (X.Num, X.Par, X.Uni))
        assert not X.collect(u, X.Ref) and not X.collect(u, X.GRef) and not X.collect(u, X.Aux)


def test_sign_variants_share_an_entry():
    c, nu = X.Par("c", 0, "c0"), X.Par("s", 0, "nu")
    u = X.Ref(0)
    tab = X.UniformTable()
    e = X.hoist_uniforms(X.add(X.mul(X.Num(8.0), c, nu, u), X.mul(X.Num(-8.0), c, nu, X.Ref(0, 0, (0, 0, 1)))), tab)
    assert len(tab.exprs) == 1
    assert len(X.collect(e, X.Uni)) == 1


@pytest.mark.parametrize("name", sorted(SPECS))
def test_carried_values_keep_the_adjoint(name):
    """Subexpressions handed from the residual code to the same-frame adjoint (expr.find_carries): every carried tree occurs
    in the residuals, reads no scaled residual, and substituting the carried values leaves the adjoint's value unchanged."""
    s = SPECS[name]
    random.seed(hash(name) & 0xfff)
    tab = X.UniformTable()
    res = [X.hoist_uniforms(e, tab) for e in s.residuals]
    space = [X.hoist_uniforms(split_adjoint(e)[0], tab) for e in s.adjoint]
    carries = X.find_carries(res, space, limit=max(s.carry_max, 4))
    have = X.subtree_keys(res)
    for c in carries:
        assert c.key() in have and not X.collect(c, X.GRef) and X.op_count(c) >= 2 and not X.is_uniform(c)
    table = {c.key(): X.Car(i) for i, c in enumerate(carries)}
    swapped = [X.replace_subtrees(e, table) for e in space]
    if carries:
        assert any(X.collect(e, X.Car) for e in swapped)

    def value(e, env, cvals):
        if isinstance(e, X.Car):
            return cvals[e.i]
        if isinstance(e, (X.Num, X.Uni, X.Par, X.Ref, X.Aux, X.GRef)):
            return _value(e, env, tab)
        if isinstance(e, X.Add):
            return sum(value(t, env, cvals) for t in e.terms)
        if isinstance(e, X.Mul):
            out = 1.0
            for t in e.factors:
                out *= value(t, env, cvals)
            return out
        if isinstance(e, X.Recip):
            return 1.0 / value(e.e, env, cvals)
        if isinstance(e, X.Sel):
            return value(e.a, env, cvals) if value(e.c, env, cvals) >= 0 else value(e.b, env, cvals)
        if isinstance(e, X.PSel):
            return value(e.a, env, cvals) if env.setdefault(e.p.key(), random.random() < 0.5) else value(e.b, env, cvals)
        raise TypeError(type(e))

    for trial in range(3):
        env = {}
        cvals = [_value(c, env, tab) for c in carries]
        for e, w in zip(space, swapped):
            a, b = _value(e, env, tab), value(w, env, cvals)
            assert abs(a - b) <= 1e-9 * max(1.0, abs(a)), (name, a, b)


def test_taylor_green_carries_its_four_derivatives():
    from pde_amd.stencil.codegen import emit_spec
    txt = emit_spec(SPECS["tg2d"])
    assert "static constexpr int NCAR = 4;" in txt and "residual_c" in txt and "adjoint_space_c" in txt


@pytest.mark.parametrize("name", sorted(SPECS))
def test_shared_products_keep_their_value(name):
    """Regrouping products around sub-products that occur elsewhere (expr.share_products) changes no value."""
    s = SPECS[name]
    random.seed(hash(name) & 0xff)
    tab = X.UniformTable()
    res = [X.hoist_uniforms(e, tab) for e in s.residuals]
    shared = X.share_products(res)
    for trial in range(3):
        env = {}
        for e, h in zip(res, shared):
            a, b = _value(e, env, tab), _value(h, env, tab)
            assert abs(a - b) <= 1e-9 * max(1.0, abs(a)), (name, a, b)
