"""The single-kernel attention of the long-sequence level (csrc/attn_flash.hip: T >= 1024, C = 384 -- ObsNet's ds = 4 level, 32 x 64 = 2048 keys at
3x128x256, 32 x 32 at the config shape): AttentionBlock (openaimodel.py:278-333 over QKVAttentionLegacy :365-381) against the reference golden and
against the CPU oracle, in every arithmetic mode that routes there (f16x3, f16mx -- whose attention is f16x3 -- and the reduced-precision f16), next
to exact fp32 (three-launch path) on the same inputs; large / tiny input ranges (the per-image power-of-two guard of q, k, v); batches that do and do
not fill whole XCD groups; rows of a batch equal their own single runs."""
import pytest
import torch

from conftest import gold, rel_l2
from drmnet_amd import ops, synth
from oracle import unet as ou
from test_gpu_ops import attn_manifest, block_inputs

pytestmark = pytest.mark.gpu
TOL = {"fp32": 2e-5, "f16x3": 2e-5, "f16mx": 2e-5, "f16": 5e-3, "bf16": 3e-2}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no fallback)"
    return torch.device("cuda:0")


def run(P, x, precision, dev):
    try:
        ops.set_precision(precision)
        return ops.attention_block([p.to(dev) for p in P.values()], x.to(dev)).cpu()
    finally:
        ops.set_precision("fp32")


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16mx", "f16", "bf16"])
def test_attention_block_384_at_32x32_vs_reference_golden(dev, precision):
    gd = gold("attnblock_384_32x32")
    x, _ = block_inputs(384, 384, 32, 32, int(gd["n"]))
    P = synth.synth_state_dict(attn_manifest(384), int(gd["seed"]))
    err = rel_l2(run(P, x, precision, dev), gd["out"])
    print(f"attention 384 @32x32 ({precision}): rel-L2 vs the reference {err:.2e}")
    assert err < TOL[precision]


@pytest.mark.parametrize("precision", ["f16x3", "f16", "bf16"])
@pytest.mark.parametrize("n,h,w", [(3, 32, 64), (8, 32, 32), (5, 32, 32)])
def test_attention_block_vs_oracle_metric_shape(dev, n, h, w, precision):
    """T = 2048 (the 3x128x256 metric shape's ds = 4 level) and T = 1024; N * T / 128 a multiple of 8 (XCD-grouped query tiles) and not"""
    gen = torch.Generator().manual_seed(100 + n + h + w)
    x = torch.randn((n, 384, h, w), generator=gen)
    x[0] *= 3.0  # images of different scale in one batch: the per-image factors of q / k / v differ
    P = synth.synth_state_dict(attn_manifest(384), 11)
    ref = ou.attention_block({"ab." + k: v for k, v in P.items()}, ou.Attn("ab", 384), x)
    out = run(P, x, precision, dev)
    e_all = rel_l2(out, ref)
    e_branch = rel_l2(out - x, ref - x.double())  # the residual x is added exactly: the attention branch itself
    print(f"attention 384 @{h}x{w} N={n} ({precision}): {e_all:.2e}, branch {e_branch:.2e}")
    assert torch.isfinite(out).all() and e_all < TOL[precision] and e_branch < {"f16x3": 1e-4, "f16": 2e-2, "bf16": 1e-1}[precision]
    # a row of the batch == that image alone (per-image factors, no cross-image state)
    one = run(P, x[1:2].contiguous(), precision, dev)
    # (f16: the qkv conv in front runs another tile family at N = 1 -- its sums land on the other side of an fp16 rounding boundary for a few operands)
    assert rel_l2(one[0], out[1]) < {"f16x3": 1e-6, "f16": 1e-4, "bf16": 1e-3}[precision]


@pytest.mark.parametrize("vscale,qscale", [(3e4, 1.0), (1e-3, 30.0), (30.0, 1.0 / 30.0)])
def test_input_ranges_beyond_fp16(dev, vscale, qscale):
    """v rows far above fp16's limit, or q / k / v up to 3e4 apart from each other (the scores themselves unchanged): each of the three operands
    is staged through its own per-image power of two.  (The qkv conv in front packs its weight tensor with ONE power of two, so the row groups
    of qkv.weight are kept within the 2^-24 .. 1 window of its fp16 hi + lo image: 3e4 apart at most.)  Oracle in fp64."""
    gen = torch.Generator().manual_seed(7)
    x = torch.randn((2, 384, 32, 32), generator=gen)
    P = synth.synth_state_dict(attn_manifest(384), 12)
    C = 384
    w, b = P["qkv.weight"].clone(), P["qkv.bias"].clone()
    for lo, f in ((0, qscale), (C, 1.0 / qscale), (2 * C, vscale)):
        w[lo:lo + C] *= f
        b[lo:lo + C] *= f
    P["qkv.weight"], P["qkv.bias"] = w, b
    P["proj_out.weight"] = P["proj_out.weight"] / vscale
    with ou.working_dtype(torch.float64):
        ref = ou.attention_block({"ab." + k: v.double() for k, v in P.items()}, ou.Attn("ab", 384), x.double())
    out = run(P, x, "f16x3", dev)
    e = rel_l2(out - x, ref - x.double())
    print(f"attention branch with v x {vscale:g}, q x {qscale:g}, k / {qscale:g}: rel-L2 vs the fp64 oracle {e:.2e}")
    assert torch.isfinite(out).all() and e < 1e-4
