"""Tiny symbolic engine for finite-difference residuals on a structured grid.

A residual is an expression tree over *relative* grid references.  The engine
differentiates a residual with respect to each referenced grid value and
re-centres the derivative so that a *gather* adjoint stencil can be emitted:

    grad_f(q) = sum_k sum_{refs r=(f,ot,o) of R_k}  G_k(q - o, t - ot) * dR_k/dr |_(q - o, t - ot)

Nodes
-----
Num(v)                 literal
Par(kind, i, name)     runtime scalar: kind 'c' per-instance constant, 's' per-sample parameter
Ref(f, ot, o)          field f at time offset ot and spatial offset o=(o0,o1,o2) (tensor dims -3,-2,-1)
Aux(a, o)              time-independent auxiliary field a at spatial offset o (never differentiated)
GRef(k, ot, o)         scaled residual G_k = coef_k * R_k at a relative site (adjoint expressions only)
Uni(i)                 entry i of a table of hoisted uniform subexpressions (emission only, see hoist_uniforms)
Add(terms), Mul(factors), Recip(e)
Sel(c, a, b)           c >= 0 ? a : b      (derivative flows through the selected branch only)
PSel(p, a, b)          position predicate p ? a : b, p = Pred(axis, k, hi): index == k  (hi: == N-1-k)
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Tuple

O3 = Tuple[int, int, int]
ZERO3: O3 = (0, 0, 0)


def addo(a: O3, b: O3) -> O3:
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def nego(a: O3) -> O3:
    return (-a[0], -a[1], -a[2])


class E:
    __slots__ = ("_key",)

    def key(self):
        raise NotImplementedError

    def __hash__(self):
        return hash(self.key())

    def __eq__(self, other):
        return isinstance(other, E) and self.key() == other.key()

    # arithmetic sugar
    def __add__(self, o):
        return add(self, wrap(o))

    def __radd__(self, o):
        return add(wrap(o), self)

    def __sub__(self, o):
        return add(self, mul(Num(-1.0), wrap(o)))

    def __rsub__(self, o):
        return add(wrap(o), mul(Num(-1.0), self))

    def __mul__(self, o):
        return mul(self, wrap(o))

    def __rmul__(self, o):
        return mul(wrap(o), self)

    def __neg__(self):
        return mul(Num(-1.0), self)

    def __truediv__(self, o):
        o = wrap(o)
        if isinstance(o, Num):
            return mul(self, Num(1.0 / o.v))
        return mul(self, recip(o))


class Num(E):
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = float(v)

    def key(self):
        return ("n", self.v)


class Par(E):
    __slots__ = ("kind", "i", "name")

    def __init__(self, kind: str, i: int, name: str):
        self.kind, self.i, self.name = kind, i, name

    def key(self):
        return ("p", self.kind, self.i)


class Ref(E):
    __slots__ = ("f", "ot", "o")

    def __init__(self, f: int, ot: int = 0, o: O3 = ZERO3):
        self.f, self.ot, self.o = f, ot, tuple(o)

    def key(self):
        return ("r", self.f, self.ot, self.o)


class Aux(E):
    __slots__ = ("a", "o")

    def __init__(self, a: int, o: O3 = ZERO3):
        self.a, self.o = a, tuple(o)

    def key(self):
        return ("x", self.a, self.o)


class GRef(E):
    __slots__ = ("k", "ot", "o")

    def __init__(self, k: int, ot: int, o: O3):
        self.k, self.ot, self.o = k, ot, tuple(o)

    def key(self):
        return ("g", self.k, self.ot, self.o)


class Uni(E):
    __slots__ = ("i",)

    def __init__(self, i: int):
        self.i = i

    def key(self):
        return ("u", self.i)


class Car(E):
    """Value number i handed from the residual code to the adjoint code of the same site and frame (emission only)."""
    __slots__ = ("i",)

    def __init__(self, i: int):
        self.i = i

    def key(self):
        return ("car", self.i)


class Add(E):
    __slots__ = ("terms",)

    def __init__(self, terms):
        self.terms = tuple(terms)

    def key(self):
        return ("+",) + tuple(t.key() for t in self.terms)


class Mul(E):
    __slots__ = ("factors",)

    def __init__(self, factors):
        self.factors = tuple(factors)

    def key(self):
        return ("*",) + tuple(t.key() for t in self.factors)


class Recip(E):
    __slots__ = ("e",)

    def __init__(self, e):
        self.e = e

    def key(self):
        return ("1/", self.e.key())


class Pred:
    """index along `axis` equals k (hi=False) or N-1-k (hi=True)."""
    __slots__ = ("axis", "k", "hi")

    def __init__(self, axis: int, k: int, hi: bool = False):
        self.axis, self.k, self.hi = axis, k, hi

    def key(self):
        return ("P", self.axis, self.k, self.hi)

    def shifted(self, do: O3) -> "Pred":
        # predicate held at site s = q + do; rewrite it for q
        d = do[self.axis]
        return Pred(self.axis, self.k + d if self.hi else self.k - d, self.hi)


class Sel(E):
    __slots__ = ("c", "a", "b")

    def __init__(self, c, a, b):
        self.c, self.a, self.b = c, a, b

    def key(self):
        return ("?", self.c.key(), self.a.key(), self.b.key())


class PSel(E):
    __slots__ = ("p", "a", "b")

    def __init__(self, p: Pred, a, b):
        self.p, self.a, self.b = p, a, b

    def key(self):
        return ("P?", self.p.key(), self.a.key(), self.b.key())


ZERO = Num(0.0)
ONE = Num(1.0)


def wrap(x) -> E:
    return x if isinstance(x, E) else Num(x)


def is_zero(e: E) -> bool:
    return isinstance(e, Num) and e.v == 0.0


def add(*xs: E) -> E:
    terms: List[E] = []
    c = 0.0
    for x in xs:
        if isinstance(x, Add):
            sub = x.terms
        else:
            sub = (x,)
        for t in sub:
            if isinstance(t, Num):
                c += t.v
            else:
                terms.append(t)
    if c != 0.0:
        terms.append(Num(c))
    if not terms:
        return ZERO
    if len(terms) == 1:
        return terms[0]
    return Add(terms)


def mul(*xs: E) -> E:
    facs: List[E] = []
    c = 1.0
    for x in xs:
        sub = x.factors if isinstance(x, Mul) else (x,)
        for t in sub:
            if isinstance(t, Num):
                c *= t.v
            else:
                facs.append(t)
    if c == 0.0:
        return ZERO
    if not facs:
        return Num(c)
    if c != 1.0:
        facs.insert(0, Num(c))
    if len(facs) == 1:
        return facs[0]
    return Mul(facs)


def recip(e: E) -> E:
    if isinstance(e, Num):
        return Num(1.0 / e.v)
    return Recip(e)


def sel(c: E, a: E, b: E) -> E:
    a, b = wrap(a), wrap(b)
    if a == b:
        return a
    return Sel(c, a, b)


def psel(p: Pred, a: E, b: E) -> E:
    a, b = wrap(a), wrap(b)
    if a == b:
        return a
    return PSel(p, a, b)


# ---------------------------------------------------------------------------
# traversal
# ---------------------------------------------------------------------------

def children(e: E) -> Iterable[E]:
    if isinstance(e, Add):
        return e.terms
    if isinstance(e, Mul):
        return e.factors
    if isinstance(e, Recip):
        return (e.e,)
    if isinstance(e, Sel):
        return (e.c, e.a, e.b)
    if isinstance(e, PSel):
        return (e.a, e.b)
    return ()


def collect(e: E, cls, out=None, seen=None):
    """All distinct nodes of type cls inside e (conditions included)."""
    if out is None:
        out, seen = {}, set()
    k = e.key()
    if k in seen:
        return out
    seen.add(k)
    if isinstance(e, cls):
        out[k] = e
    for ch in children(e):
        collect(ch, cls, out, seen)
    return out


def diff(e: E, r: Ref, memo=None) -> E:
    """d e / d r, with no derivative through Sel conditions."""
    if memo is None:
        memo = {}
    k = e.key()
    if k in memo:
        return memo[k]
    if isinstance(e, Ref):
        res = ONE if e == r else ZERO
    elif isinstance(e, (Num, Par, Uni, Aux, GRef, Car)):
        res = ZERO
    elif isinstance(e, Add):
        res = add(*[diff(t, r, memo) for t in e.terms])
    elif isinstance(e, Mul):
        parts = []
        for i, f in enumerate(e.factors):
            d = diff(f, r, memo)
            if is_zero(d):
                continue
            parts.append(mul(d, *[g for j, g in enumerate(e.factors) if j != i]))
        res = add(*parts) if parts else ZERO
    elif isinstance(e, Recip):
        d = diff(e.e, r, memo)
        res = ZERO if is_zero(d) else mul(Num(-1.0), e, e, d)
    elif isinstance(e, Sel):
        res = sel(e.c, diff(e.a, r, memo), diff(e.b, r, memo))
    elif isinstance(e, PSel):
        res = psel(e.p, diff(e.a, r, memo), diff(e.b, r, memo))
    else:
        raise TypeError(type(e))
    memo[k] = res
    return res


def shift(e: E, dt: int, do: O3, memo=None) -> E:
    """Re-express e (held at site s, time t) relative to q where s = q + do, t = tq + dt."""
    if memo is None:
        memo = {}
    k = e.key()
    if k in memo:
        return memo[k]
    if isinstance(e, Ref):
        res = Ref(e.f, e.ot + dt, addo(e.o, do))
    elif isinstance(e, Aux):
        res = Aux(e.a, addo(e.o, do))
    elif isinstance(e, GRef):
        res = GRef(e.k, e.ot + dt, addo(e.o, do))
    elif isinstance(e, (Num, Par, Uni, Car)):
        res = e
    elif isinstance(e, Add):
        res = add(*[shift(t, dt, do, memo) for t in e.terms])
    elif isinstance(e, Mul):
        res = mul(*[shift(t, dt, do, memo) for t in e.factors])
    elif isinstance(e, Recip):
        res = recip(shift(e.e, dt, do, memo))
    elif isinstance(e, Sel):
        res = sel(shift(e.c, dt, do, memo), shift(e.a, dt, do, memo), shift(e.b, dt, do, memo))
    elif isinstance(e, PSel):
        res = psel(e.p.shifted(do), shift(e.a, dt, do, memo), shift(e.b, dt, do, memo))
    else:
        raise TypeError(type(e))
    memo[k] = res
    return res


def adjoint_exprs(residuals: List[E], nfields: int) -> List[E]:
    """Gather-form adjoint: one expression per field, over GRef and Ref nodes relative to the output site."""
    out: List[List[E]] = [[] for _ in range(nfields)]
    for k, R in enumerate(residuals):
        refs = collect(R, Ref)
        for rk in sorted(refs):
            r = refs[rk]
            J = diff(R, r)
            if is_zero(J):
                continue
            do = nego(r.o)
            out[r.f].append(mul(GRef(k, -r.ot, do), shift(J, -r.ot, do)))
    return [add(*ts) if ts else ZERO for ts in out]


# ---------------------------------------------------------------------------
# footprint analysis
# ---------------------------------------------------------------------------

def footprint(exprs: List[E], cls) -> Dict[str, object]:
    """Per-axis max |offset| and time-offset range of nodes of type cls, split by ot==0 / ot!=0."""
    h0 = [0, 0, 0]
    ht = [0, 0, 0]
    tmin, tmax = 0, 0
    n = 0
    for e in exprs:
        for node in collect(e, cls).values():
            n += 1
            ot = getattr(node, "ot", 0)
            tmin, tmax = min(tmin, ot), max(tmax, ot)
            tgt = h0 if ot == 0 else ht
            for a in range(3):
                tgt[a] = max(tgt[a], abs(node.o[a]))
    return {"h0": tuple(h0), "ht": tuple(ht), "tmin": tmin, "tmax": tmax, "count": n}


# ---------------------------------------------------------------------------
# uniform hoisting
# ---------------------------------------------------------------------------
# Products and sums of literals, constants and per-sample parameters ("8 * c1", "16 * c4 * nu") are the same
# at every grid point of a sample.  Left inside the per-point code they cost vector instructions at every
# point (the device compiler keeps float arithmetic in source order and cannot fold "8 * t * c1" into one
# multiply by a precomputed "8 c1").  hoist_uniforms() rewrites expressions so that every maximal uniform
# subexpression, and the uniform part of every mixed product, becomes a reference Uni(i) into a table that the
# kernels evaluate once per sample.  Values that differ only in sign share an entry.

def is_uniform(e: E) -> bool:
    if isinstance(e, (Num, Par, Uni)):
        return True
    if isinstance(e, (Add, Mul, Recip)):
        return all(is_uniform(c) for c in children(e))
    return False


class UniformTable:
    def __init__(self):
        self.exprs: List[E] = []
        self.index: Dict[object, int] = {}

    def ref(self, e: E) -> E:
        """Uni node (possibly negated) standing for the uniform, non-atomic expression e."""
        if isinstance(e, Mul) and isinstance(e.factors[0], Num) and e.factors[0].v < 0:
            pos = mul(Num(-e.factors[0].v), *e.factors[1:])
            return mul(Num(-1.0), pos if isinstance(pos, (Num, Par)) else self.ref(pos))
        k = e.key()
        if k not in self.index:
            self.index[k] = len(self.exprs)
            self.exprs.append(e)
        return Uni(self.index[k])


def hoist_uniforms(e: E, tab: UniformTable, memo=None) -> E:
    if memo is None:
        memo = {}
    k = e.key()
    if k in memo:
        return memo[k]
    if isinstance(e, (Num, Par, Uni, Ref, Aux, GRef, Car)):
        res = e
    elif is_uniform(e):
        res = tab.ref(e)
    elif isinstance(e, Add):
        us = [t for t in e.terms if is_uniform(t)]
        vs = [hoist_uniforms(t, tab, memo) for t in e.terms if not is_uniform(t)]
        if us:
            u = add(*us)
            vs.append(u if isinstance(u, (Num, Par)) else tab.ref(u))
        res = add(*vs)
    elif isinstance(e, Mul):
        us = [f for f in e.factors if is_uniform(f)]
        vs = [hoist_uniforms(f, tab, memo) for f in e.factors if not is_uniform(f)]
        if us:
            u = mul(*us)
            vs.insert(0, u if isinstance(u, (Num, Par)) else tab.ref(u))
        res = mul(*vs)
    elif isinstance(e, Recip):
        res = recip(hoist_uniforms(e.e, tab, memo))
    elif isinstance(e, Sel):
        res = sel(hoist_uniforms(e.c, tab, memo), hoist_uniforms(e.a, tab, memo), hoist_uniforms(e.b, tab, memo))
    elif isinstance(e, PSel):
        res = psel(e.p, hoist_uniforms(e.a, tab, memo), hoist_uniforms(e.b, tab, memo))
    else:
        raise TypeError(type(e))
    memo[k] = res
    return res


# ---------------------------------------------------------------------------
# values shared by the residual and the adjoint of one site
# ---------------------------------------------------------------------------
# The recomputing adjoint evaluates the residuals and, one barrier later, the same-frame adjoint at the lane's own
# points.  Subexpressions of field values that both contain (a derivative at the site, an upwind face value) can be
# handed over in registers instead of being recomputed from LDS.

def subtree_keys(exprs: Iterable[E]) -> set:
    out: set = set()

    def walk(e: E):
        k = e.key()
        if k in out:
            return
        out.add(k)
        for ch in children(e):
            walk(ch)

    for e in exprs:
        walk(e)
    return out


def op_count(e: E) -> int:
    """Arithmetic operations of a tree (shared subtrees counted once per occurrence)."""
    if isinstance(e, Add):
        return len(e.terms) - 1 + sum(op_count(t) for t in e.terms)
    if isinstance(e, Mul):
        return len(e.factors) - 1 + sum(op_count(t) for t in e.factors)
    if isinstance(e, Recip):
        return 4 + op_count(e.e)
    if isinstance(e, (Sel, PSel)):
        return 1 + sum(op_count(c) for c in children(e))
    return 0


def find_carries(residuals: List[E], adjoint: List[E], min_ops: int = 2, limit: int = 8) -> List[E]:
    """Maximal non-uniform subtrees of the adjoint expressions that also occur in the residuals, most expensive first."""
    have = subtree_keys(residuals)
    found: Dict[object, E] = {}

    def walk(e: E):
        if isinstance(e, (Num, Par, Uni, Ref, Aux, GRef, Car)) or is_uniform(e):
            return
        if e.key() in have and op_count(e) >= min_ops and not collect(e, GRef):
            found.setdefault(e.key(), e)
            return
        for ch in children(e):
            walk(ch)

    for e in adjoint:
        walk(e)
    return sorted(found.values(), key=lambda x: -op_count(x))[:limit]


def share_products(exprs: List[E]) -> List[E]:
    """Products that contain another product of the same expressions as a sub-product are regrouped around it
    (u * c * d  with  c * d  elsewhere  ->  u * (c * d)), so that the emitter's temporaries share the multiplication."""
    muls: Dict[object, Mul] = {}

    def walk(e: E, seen: set):
        k = e.key()
        if k in seen:
            return
        seen.add(k)
        if isinstance(e, Mul):
            muls[k] = e
        for ch in children(e):
            walk(ch, seen)

    seen: set = set()
    for e in exprs:
        walk(e, seen)
    small = sorted((m for m in muls.values() if len(m.factors) >= 2), key=lambda m: -len(m.factors))

    def regroup(e: E, memo: Dict[object, E]) -> E:
        k = e.key()
        if k in memo:
            return memo[k]
        if isinstance(e, Add):
            res = Add([regroup(t, memo) for t in e.terms])
        elif isinstance(e, Mul):
            facs = list(e.factors)
            keys = [f.key() for f in facs]
            inner = None
            if len(facs) >= 3:
                for m in small:
                    if len(m.factors) >= len(facs) or m.key() == k:
                        continue
                    left = list(keys)
                    ok = True
                    for f in m.factors:
                        if f.key() in left:
                            left.remove(f.key())
                        else:
                            ok = False
                            break
                    if ok:
                        inner = m
                        rest = []
                        need = [f.key() for f in m.factors]
                        for f in facs:
                            if f.key() in need:
                                need.remove(f.key())
                            else:
                                rest.append(f)
                        facs = rest + [m]
                        break
            res = Mul([f if f is inner else regroup(f, memo) for f in facs])
        elif isinstance(e, Recip):
            res = Recip(regroup(e.e, memo))
        elif isinstance(e, Sel):
            res = Sel(regroup(e.c, memo), regroup(e.a, memo), regroup(e.b, memo))
        elif isinstance(e, PSel):
            res = PSel(e.p, regroup(e.a, memo), regroup(e.b, memo))
        else:
            res = e
        memo[k] = res
        return res

    memo: Dict[object, E] = {}
    return [regroup(e, memo) for e in exprs]


# ---------------------------------------------------------------------------
# adjoint regrouped around the field values at the output point
# ---------------------------------------------------------------------------
# adjoint_exprs() yields one term per tap of a scaled residual, each with its own coefficient expression:
#     G(+1) * (-(8 c u) - l)  +  G(-1) * ((8 c u) - l)  +  G(+2) * (c u)  - ...
# Where the coefficients are polynomials in field values AT THE OUTPUT POINT (conservative products such as d(phi u)/dx),
# the same sum ordered by those field values is a few plain stencils of G with uniform weights, each scaled once:
#     u * (c G(+2) - 8c G(+1) + 8c G(-1) - c G(-2))  +  (-l G(+1) - l G(-1) + ...)
# i.e. one multiply-add per tap with a hoisted uniform weight, instead of two or three operations per tap.

def _monomials(e: E, limit: int = 4096):
    """e as a list of (number, tuple of atomic factors) whose sum of products equals e; None when the expansion would
    exceed `limit` monomials.  Everything but Add, Mul and Num is atomic."""
    if isinstance(e, Num):
        return [(e.v, ())]
    if isinstance(e, Add):
        out = []
        for t in e.terms:
            m = _monomials(t, limit)
            if m is None:
                return None
            out += m
            if len(out) > limit:
                return None
        return out
    if isinstance(e, Mul):
        out = [(1.0, ())]
        for f in e.factors:
            m = _monomials(f, limit)
            if m is None or len(out) * len(m) > limit:
                return None
            out = [(c1 * c2, a1 + a2) for c1, a1 in out for c2, a2 in m]
        return out
    return [(1.0, (e,))]


def factor_pointwise(e: E) -> E:
    """The adjoint expression e regrouped as  sum over products P of output-point values  P * (sum_i w_i G_i)  with uniform
    weights w_i; e itself when it does not have that shape (a term without exactly one scaled residual, a field value
    away from the output point or frame, an expansion that explodes) or when the regrouped form has no fewer operations."""
    mons = _monomials(e)
    if mons is None:
        return e
    groups: Dict[object, Dict[object, list]] = {}
    order: List[object] = []
    points: Dict[object, Tuple[E, ...]] = {}
    for c, atoms in mons:
        if c == 0.0:
            continue
        gs = [a for a in atoms if isinstance(a, GRef)]
        if len(gs) != 1:
            return e
        uni = sorted((a for a in atoms if isinstance(a, (Par, Uni))), key=lambda a: a.key())
        pts = sorted((a for a in atoms if not isinstance(a, (Par, Uni, GRef))), key=lambda a: a.key())
        for a in pts:      # only values at the output point and frame may be pulled out of the tap sums
            for r in collect(a, Ref).values():
                if r.ot != 0 or r.o != ZERO3:
                    return e
            if collect(a, GRef):
                return e
        pk = tuple(a.key() for a in pts)
        if pk not in groups:
            groups[pk] = {}
            order.append(pk)
            points[pk] = tuple(pts)
        slot = groups[pk].setdefault((gs[0].key(), tuple(a.key() for a in uni)), [0.0, gs[0], uni])
        slot[0] += c
    terms = []
    for pk in order:
        taps = [mul(Num(c), *uni, g) for c, g, uni in groups[pk].values() if c != 0.0]
        if taps:
            terms.append(mul(*points[pk], add(*taps)))
    res = add(*terms) if terms else ZERO
    return res if op_count(res) < op_count(e) else e


def replace_subtrees(e: E, table: Dict[object, E], memo=None) -> E:
    if memo is None:
        memo = {}
    k = e.key()
    if k in memo:
        return memo[k]
    if k in table:
        res = table[k]
    elif isinstance(e, Add):
        res = add(*[replace_subtrees(t, table, memo) for t in e.terms])
    elif isinstance(e, Mul):
        res = mul(*[replace_subtrees(t, table, memo) for t in e.factors])
    elif isinstance(e, Recip):
        res = recip(replace_subtrees(e.e, table, memo))
    elif isinstance(e, Sel):
        res = sel(replace_subtrees(e.c, table, memo), replace_subtrees(e.a, table, memo), replace_subtrees(e.b, table, memo))
    elif isinstance(e, PSel):
        res = psel(e.p, replace_subtrees(e.a, table, memo), replace_subtrees(e.b, table, memo))
    else:
        res = e
    memo[k] = res
    return res


# ---------------------------------------------------------------------------
# C emission with common-subexpression temporaries
# ---------------------------------------------------------------------------

def _flt(v: float) -> str:
    """A literal in the scalar type of the accessor (R_ = typename A::real: float in the training kernels, double in the
    fp64 residual kernel), written with all the digits of the double: the float kernels get its nearest float."""
    s = repr(float(v))
    if "e" not in s and "E" not in s and "." not in s and "inf" not in s and "nan" not in s:
        s += ".0"
    return f"R_({s})"


class Emitter:
    """Prints expressions as C with temporaries for shared non-trivial subtrees."""

    def __init__(self, acc: str = "a"):
        self.acc = acc
        self.lines: List[str] = []
        self.names: Dict[object, str] = {}
        self.count: Dict[object, int] = {}

    def _count(self, e: E):
        k = e.key()
        self.count[k] = self.count.get(k, 0) + 1
        if self.count[k] == 1:
            for ch in children(e):
                self._count(ch)

    def prepare(self, exprs: Iterable[E]):
        for e in exprs:
            self._count(e)

    def _atom(self, e: E) -> str:
        a = self.acc
        if isinstance(e, Num):
            # 0, 0.5, 1, 2, 4 are encoded inside a VALU instruction; any other literal goes through the accessor,
            # which may keep it in a vector register (a literal in a 3-operand instruction costs a scalar operand)
            return _flt(e.v) if abs(e.v) in (0.0, 0.5, 1.0, 2.0, 4.0) else f"{a}.k({_flt(e.v)})"
        if isinstance(e, Par):
            return f"{a}.{'c' if e.kind == 'c' else 'sp'}({e.i})"
        if isinstance(e, Uni):
            return f"{a}.u({e.i})"
        if isinstance(e, Ref):
            return f"{a}.F({e.f},{e.ot},{e.o[0]},{e.o[1]},{e.o[2]})"
        if isinstance(e, Aux):
            return f"{a}.X({e.a},{e.o[0]},{e.o[1]},{e.o[2]})"
        if isinstance(e, GRef):
            return f"{a}.G({e.k},{e.ot},{e.o[0]},{e.o[1]},{e.o[2]})"
        if isinstance(e, Car):
            return f"C[{e.i}]"
        raise TypeError(type(e))

    def expr(self, e: E) -> str:
        k = e.key()
        if k in self.names:
            return self.names[k]
        if isinstance(e, (Num, Par, Uni, Car)):
            return self._atom(e)
        if isinstance(e, (Ref, Aux, GRef)):
            s = self._atom(e)
        elif isinstance(e, Add):
            parts = []
            for i, t in enumerate(e.terms):
                neg = False
                if isinstance(t, Mul) and isinstance(t.factors[0], Num) and t.factors[0].v < 0:
                    neg = True
                    t = mul(Num(-t.factors[0].v), *t.factors[1:])
                elif isinstance(t, Num) and t.v < 0:
                    neg = True
                    t = Num(-t.v)
                s_t = self.expr(t)
                if i == 0:
                    parts.append(("-" if neg else "") + s_t)
                else:
                    parts.append((" - " if neg else " + ") + s_t)
            s = "(" + "".join(parts) + ")"
        elif isinstance(e, Mul):
            s = "(" + " * ".join(self.expr(t) for t in e.factors) + ")"
        elif isinstance(e, Recip):
            s = "(R_(1.0) / " + self.expr(e.e) + ")"
        elif isinstance(e, Sel):
            # a call, so that both sides are values before the choice: the device compiler then emits one select;
            # with the ?: operator it builds divergent branches around the few multiply-adds of each side
            s = f"{self.acc}.sel({self.expr(e.c)}, {self.expr(e.a)}, {self.expr(e.b)})"
        elif isinstance(e, PSel):
            fn = "ath" if e.p.hi else "at"
            s = f"{self.acc}.pick({self.acc}.{fn}({e.p.axis},{e.p.k}), {self.expr(e.a)}, {self.expr(e.b)})"
        else:
            raise TypeError(type(e))
        # loads and shared subtrees become named temporaries
        if isinstance(e, (Ref, Aux, GRef)) or self.count.get(k, 0) > 1:
            name = f"t{len(self.names)}"
            self.lines.append(f"const R_ {name} = {s};")
            self.names[k] = name
            return name
        return s
