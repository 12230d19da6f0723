"""CPU pin of the arithmetic claim behind the ring kernels (no GPU): the numpy model in oracle/split_model.py restates the
operand cuts of conv_gemm.hip; here its error against float64 is held to the figures DESIGN.md section 4 quotes from the
GPU lab (profiles/ring_lab_r03_range.txt), including the failure modes the amax scale exists for."""
import numpy as np
import pytest

from oracle import split_model as S


def _operands(seed, m=96, k=2048, n=64, spread=7):
    rng = np.random.default_rng(seed)
    def draw(shape):
        return (rng.uniform(-1, 1, shape) * 2.0 ** (-rng.integers(0, spread + 1, shape))).astype(np.float32)
    return draw((m, k)), draw((k, n))


def _err(c, a, b):
    ref = a.astype(np.float64) @ b.astype(np.float64)
    return float(np.linalg.norm(c - ref) / np.linalg.norm(ref))


def test_scale_exponent_rule():
    for amax in (1e-30, 3.7e-9, 7.3e-4, 0.02, 1.0, 1.999, 2.0, 117.0, 6.5e4, 1e30):
        for top in (S.PC_TOP_A, S.PC_TOP_B):
            k = S.scale_exp(amax, top)
            if abs(k) < 100:
                assert 2.0 ** (top - 1) <= np.float32(amax) * 2.0 ** k < 2.0 ** top, (amax, top, k)
    assert S.scale_exp(0.0, 12) == 0 and S.scale_exp(np.inf, 12) == 0 and S.scale_exp(np.nan, 12) == 0
    assert S.scale_exp(1e-38, 12) == 100                      # clamped: 2^k stays a normal float


@pytest.mark.parametrize("spread", [7, 20, 30])
def test_two_fp16_planes_match_three_bf16_planes_and_fp32(spread):
    a, b = _operands(spread, spread=spread)
    e2, e3 = _err(S.matmul_two_plane(a, b), a, b), _err(S.matmul_three_plane(a, b), a, b)
    e32 = _err(a @ b, a, b)
    print("spread %d: two fp16 planes %.2e  three bf16 planes %.2e  plain fp32 matmul %.2e" % (spread, e2, e3, e32))
    assert e2 < 6e-7 and e3 < 8e-7                            # GPU lab: 3.9e-7 .. 4.2e-7 and 3.7e-7 .. 5.0e-7
    assert e2 < 4 * max(e32, 1e-7)


def test_gradient_magnitudes_need_the_scale_and_tolerate_a_generous_one():
    a, b = _operands(11)
    a = (a * np.float32(2.0 ** -20)).astype(np.float32)       # activations' gradients: ~1e-6
    good = _err(S.matmul_two_plane(a, b), a, b)
    unscaled = _err(S.matmul_two_plane(a, b, ka=0), a, b)
    k = S.scale_exp(np.abs(a).max(), S.PC_TOP_A)             # amax * 2^k in [2^11, 2^12)
    at_one = _err(S.matmul_two_plane(a, b, ka=k - 12), a, b)      # amax * s ~ 1: the lab's "exact" setting
    low = _err(S.matmul_two_plane(a, b, ka=k - 20), a, b)         # amax * s ~ 2^-8: the lab's "2^8 too small"
    high = _err(S.matmul_two_plane(a, b, ka=k + 3), a, b)         # amax * s < 2^15: still below fp16's 65504
    with np.errstate(over="ignore", invalid="ignore"):
        over = S.matmul_two_plane(a, b, ka=k + 5)                 # amax * s >= 2^16: h overflows
    print("A * 2^-20: kernel's scale %.2e, none %.2e, amax*s = 1 %.2e, = 2^-8 %.2e, = 2^14 %.2e" % (good, unscaled, at_one, low, high))
    assert good < 6e-7 and at_one < 6e-7 and high < 6e-7      # lab: 3.9e-7 at amax * s = 1, 3.7e-7 at 2^8
    assert unscaled > 1e-2                                    # lab: 1.2e-1 -- why the amax is taken from the very tensor
    assert 3e-6 < low < 1e-3                                  # lab: 3.7e-5
    assert not np.all(np.isfinite(over))                      # and why its top sits a factor 16 below fp16's range


def test_one_signed_operands_carry_no_bias_beyond_fp32():
    rng = np.random.default_rng(3)
    a = np.abs(rng.normal(size=(64, 9216))).astype(np.float32)        # post-ReLU activations
    b = np.abs(rng.normal(size=(9216, 32)) * 0.02).astype(np.float32)  # a positive filter
    ref = a.astype(np.float64) @ b.astype(np.float64)
    c = S.matmul_two_plane(a, b)
    bias = float(np.mean((c - ref) / ref))
    print("one-signed K=9216: rel-L2 %.2e, mean relative bias %.2e" % (_err(c, a, b), bias))
    assert _err(c, a, b) < 4e-7 and abs(bias) < 2e-7          # GPU test_ops: 1.36e-7, bias -3.5e-8
