"""ffm_field_eval -- one element-wise expression in one pass (the Foam layer's lazily evaluated field algebra, include/ffmFoam.H: dField)
-- against the chain of ffm_field_binary / _scalar / _unary calls it replaces: bit for bit, for every operator, stack depths up to four,
immediates, repeated operands and sizes that exercise the two-elements-per-thread loop's tail.  (The operators themselves are IEEE
double operations, checked against numpy.)"""
import numpy as np
import pytest

from ffm_import import ffm

pytestmark = pytest.mark.gpu

ADD, SUB, MUL, DIV, MAX, MIN, NEGSEL = range(7)
NEG, SQR, MAG, SQRT, POS0 = range(5)


@pytest.fixture(scope="module")
def ctx():
    c = ffm.Context(0)
    yield c
    c.close()


def host(t, ctx):
    ctx.sync()
    return t.cpu().numpy()


@pytest.mark.parametrize("n", [1, 63, 257, 100003, 2 * 256 * 1024 + 5])
def test_expression_equals_the_chain_of_operator_calls(ctx, n):
    rng = np.random.default_rng(n)
    a, b, c, d = [ctx.to_device(rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)) for _ in range(4)]
    ctx._ready()
    # ((a*b - c)/(|d| + 1e-3) + 0.5*a) max b   -- depth 3, two immediates, `a` and `b` used twice
    chain = ctx.field_binary(MAX, ctx.field_binary(ADD, ctx.field_binary(DIV, ctx.field_binary(SUB, ctx.field_binary(MUL, a, b), c),
                                                                         ctx.field_scalar(ADD, ctx.field_unary(MAG, d), 1e-3)),
                                                   ctx.field_scalar(MUL, a, 0.5, True)), b)
    prog = [("load", 0), ("load", 1), ("binary", MUL), ("load", 2), ("binary", SUB), ("load", 3), ("unary", MAG), ("imm", 0), ("binary", ADD),
            ("binary", DIV), ("imm", 1), ("load", 0), ("binary", MUL), ("binary", ADD), ("load", 1), ("binary", MAX)]
    fused = ctx.field_eval([a, b, c, d], [1e-3, 0.5], prog)
    assert np.array_equal(host(fused, ctx).view(np.uint64), host(chain, ctx).view(np.uint64))
    # numpy says the same (IEEE double operations, no contraction)
    A, B, Cc, D = [host(t, ctx) for t in (a, b, c, d)]
    ref = np.fmax((A * B - Cc) / (np.abs(D) + 1e-3) + 0.5 * A, B)
    assert np.array_equal(host(fused, ctx).view(np.uint64), ref.view(np.uint64))


def test_every_operator_and_a_four_deep_stack(ctx):
    n = 4099
    rng = np.random.default_rng(7)
    arrs = [ctx.to_device(rng.standard_normal(n)) for _ in range(8)]
    ctx._ready()
    H = [host(t, ctx) for t in arrs]
    # a0 - (a1*(a2 + a3/a4)) with the quotient formed first: four values on the stack before the second operator
    prog = [("load", 0), ("load", 1), ("load", 3), ("load", 4), ("binary", DIV), ("load", 2), ("binary", ADD), ("binary", MUL), ("binary", SUB)]
    r = host(ctx.field_eval(arrs[:5], [], prog), ctx)
    assert np.array_equal(r.view(np.uint64), (H[0] - H[1] * (H[3] / H[4] + H[2])).view(np.uint64))
    # all eight arrays, min / negsel / the unary operators
    prog = [("load", 5), ("unary", SQR), ("load", 6), ("unary", MAG), ("unary", SQRT), ("binary", MIN), ("load", 7), ("binary", NEGSEL),
            ("unary", NEG), ("load", 0), ("unary", POS0), ("binary", MUL), ("load", 1), ("load", 2), ("binary", MAX), ("binary", SUB),
            ("load", 3), ("load", 4), ("binary", ADD), ("binary", DIV)]
    r = host(ctx.field_eval(arrs, [], prog), ctx)
    t = np.fmin(H[5] * H[5], np.sqrt(np.abs(H[6])))
    t = -np.where(t < 0.0, H[7], t) * np.where(H[0] >= 0.0, 1.0, 0.0)
    ref = (t - np.fmax(H[1], H[2])) / (H[3] + H[4])
    assert np.array_equal(r.view(np.uint64), ref.view(np.uint64))


def test_malformed_programs_are_refused(ctx):
    a = ctx.to_device(np.ones(16)); ctx._ready()
    for arrays, imm, prog in [([a], [], [("load", 0), ("load", 0)]),                       # two values left
                              ([a], [], [("load", 0), ("binary", ADD)]),                   # operator without operands
                              ([a], [], [("load", 1)]),                                    # no such array
                              ([a], [], [("load", 0)] * 5 + [("binary", ADD)] * 4),        # five deep
                              ([a], [], [("load", 0), ("imm", 0), ("binary", ADD)]),       # no such immediate
                              ([a], [], [("load", 0), ("unary", 9)])]:
        with pytest.raises(ffm.FfmError):
            ctx.field_eval(arrays, imm, prog)
