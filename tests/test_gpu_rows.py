"""[r6, VERDICT r5 item 6a / 6b] BASELINE configs[1]'s batch with 32 DISTINCT rows, and configs[2] as written.

The full-width batch tests of the earlier rounds repeated one golden input (odd rows flipped) with one embedding / timestep, so a per-row embedding
or timestep read from the wrong row was invisible at B >= 32.  Here every row of the 32 has its own refmap, its own noised copy, its own embedding
(IllNet) or timestep (RefNet, ObsNet); tests/golden/full_rows.npz holds the reference's outputs for rows 0, 13 and 31 (tools/make_golden.py
--only full_rows, inputs regenerable from seeds), and every row of the batch must equal its own single-row run.

configs[2] as written: ObsNet DDIM-50 at B = 32 @3x128x256 in bf16 with the hipGraph-captured step -- replay == eager bit for bit, the final state
within the bf16 tolerance of an f16x3 chain."""
import numpy as np
import pytest
import torch

from conftest import NET_TOL as MODE_TOL, gold, rel_l2
from drmnet_amd import synth
from oracle import unet as ou
from test_gpu_nets import build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no fallback)"
    return torch.device("cuda:0")


def full_rows_inputs(B=32, h=128, w=256):
    """tools/make_golden.py full_rows_inputs, regenerated from the same seeds (checksums in the fixture)"""
    x = synth.synth_refmaps(B, h, w, 4242)
    g = torch.Generator().manual_seed(4243)
    xk = x + 0.025 * torch.randn(x.shape, generator=g)
    t_emb = torch.randn((B, 128), generator=g) * (0.25 + torch.arange(B, dtype=torch.float32) / 16.0)[:, None]
    t = (torch.arange(B, dtype=torch.long) * 31 + 7) % 1000
    return x, xk, t_emb, t


def inputs_checked(gd):
    x, xk, t_emb, t = full_rows_inputs()
    # (the refmaps go through exp / log10 on the host CPU: last-bit differences between machines, 1e-8 of the checksum; the draws are exact)
    assert synth.checksum(x) == pytest.approx(float(gd["xsum"]), rel=1e-6) and synth.checksum(xk) == pytest.approx(float(gd["xksum"]), rel=1e-6)
    assert synth.checksum(t_emb) == pytest.approx(float(gd["tembsum"]), rel=1e-9) and t.tolist() == gd["t"].tolist()
    return x, xk, t_emb, t


# rows of a batch against their own single-row runs: same arithmetic, other tile shapes / split-K forms and summation orders (not bitwise).
# f16x3 sees the re-ordered fp32 sums only (measured 2e-7 .. 6e-7).  f16mx re-quantises: a 1e-7 perturbation of a staged activation moves the
# e4m3 image of its lo half across a rounding boundary on ~0.3 % of the elements (quantum 2^-3.5 of |lo| <= 2^-11 |v|), each flip a 4e-5 change of
# that product -- so two runs of the same row through different tile shapes differ by about half the mode's own distance from the reference
# (measured: IllNet 1.7e-5, ObsNet 2.1e-5, DRMNet step 1.1e-5 against 2e-5 .. 4e-5 from the reference); bounded here at the whole-network bar.
SELF_TOL = {"f16x3": 2e-6, "f16mx": 5e-5}


@pytest.mark.parametrize("precision", ["f16x3", "f16mx"])
@pytest.mark.parametrize("name", ["illnet", "refnet", "obsnet"])
def test_batch32_of_distinct_rows_embeddings_and_timesteps(dev, name, precision):
    gd = gold("full_rows")
    x, xk, t_emb, t = inputs_checked(gd)
    cfg, kind, seed = {"illnet": (ou.ILLNET_CFG, "unet", synth.SEED_ILLNET), "refnet": (ou.REFNET_CFG, "encoder", synth.SEED_REFNET),
                       "obsnet": (ou.OBSNET_CFG, "unet", synth.SEED_OBSNET)}[name]
    m = build(cfg, kind, seed, dev).set_precision(precision)
    xc = torch.cat([xk, x], 1).contiguous().to(dev)
    te, tt = t_emb.to(dev), t.to(dev)
    run = (lambda xs, sl: m(xs, t_emb=te[sl])) if name == "illnet" else (lambda xs, sl: m(xs, tt[sl]))
    out = run(xc, slice(None))
    assert out.shape[0] == 32 and torch.isfinite(out).all()
    rows = [int(r) for r in gd["rows"]]
    sub = (lambda y: y) if name == "refnet" else (lambda y: y[:, ::2, ::2])
    for k, r in enumerate(rows):
        e = rel_l2(sub(out[r]).cpu(), gd[name][k])
        print(f"{name} B=32 distinct rows ({precision}) row {r} vs the reference: {e:.2e}")
        assert e < MODE_TOL[precision], (r, e)
    # rows are really distinct (a row served another row's embedding / timestep / input would pass nothing below)
    assert rel_l2(out[0].cpu(), out[1].cpu()) > 1e-2 and rel_l2(out[13].cpu(), out[31].cpu()) > 1e-2
    worst = 0.0
    for r in range(32):
        single = run(xc[r:r + 1].contiguous(), slice(r, r + 1))
        worst = max(worst, rel_l2(out[r].cpu(), single[0].cpu()))
    print(f"{name} B=32 ({precision}): worst row vs its own single-row run {worst:.2e}")
    assert worst < SELF_TOL[precision]
    # and under a permutation of the rows (inputs AND embeddings / timesteps permuted together)
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(11)).to(dev)
    outp = m(xc[perm], t_emb=te[perm]) if name == "illnet" else m(xc[perm], tt[perm])
    assert rel_l2(outp.cpu(), out[perm].cpu()) < 1e-6
    del m, out, outp
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision", ["f16x3", "f16mx"])
def test_drmnet_step_batch32_of_distinct_rows(dev, precision):
    """One reverse step (DRMNet.p_mean_variance, models/drmnet.py:752-770, reversed_k = 3) on the 32 distinct rows: rows 0 / 13 / 31 against the
    reference's step on those rows, every row against its own single-row step (z_out, hence the z-embedding IllNet sees, differs per row)."""
    from test_gpu_configs34 import full_drmnet

    gd = gold("full_rows")
    x, xk, _, _ = inputs_checked(gd)
    m = full_drmnet(dev, precision, gamma=0.9, epsilon=0.01, max_timesteps=150)
    X, XK = x.to(dev), xk.to(dev)
    k = int(gd["step_k"])
    mean, delta, z_out = m.p_mean_variance(XK, [X], [X], reversed_k=k)
    assert delta == pytest.approx(float(gd["step_delta"]))
    for j, r in enumerate(int(r) for r in gd["rows"]):
        e, ez = rel_l2(mean[r, :, ::2, ::2].cpu(), gd["step_mean"][j]), rel_l2(z_out[r].cpu(), gd["step_z_out"][j])
        print(f"DRMNet step B=32 distinct rows ({precision}) row {r}: mean {e:.2e} z_out {ez:.2e}")
        assert e < MODE_TOL[precision] and ez < MODE_TOL[precision]
    assert rel_l2(z_out[0].cpu(), z_out[1].cpu()) > 1e-3  # the rows' BRDF estimates (and so their embeddings) differ
    worst = 0.0
    for r in range(32):
        m1, _, z1 = m.p_mean_variance(XK[r:r + 1].contiguous(), [X[r:r + 1].contiguous()], [X[r:r + 1].contiguous()], reversed_k=k)
        worst = max(worst, rel_l2(mean[r].cpu(), m1[0].cpu()), rel_l2(z_out[r].cpu(), z1[0].cpu()))
    print(f"DRMNet step B=32 ({precision}): worst row vs its own single-row step {worst:.2e}")
    assert worst < SELF_TOL[precision]
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision", ["f16x3", "f16mx"])
def test_drmnet_loop_batch128_forked_row_ranges_distinct_rows(dev, precision):
    """The north-star's per-GPU batch (1024 / 8 = 128 rows) as the product runs it: drm_drmnet_sample over two reverse steps, each step's rows as two
    ranges of 64 on forked streams (drm_drmnet_set_batch_parts default) -- here with 128 DISTINCT rows and their own noise draws.  Every 8th row of
    BOTH ranges, and the rows either side of the range boundary, against the same loop run on that row alone: a row offset lost in the second range
    (inputs, timestep, z embedding, update, noise) would show there."""
    from test_gpu_configs34 import full_drmnet

    B, T = 128, 2
    x = synth.synth_refmaps(B, 128, 256, 4242)
    g = torch.Generator().manual_seed(4245)
    n0 = torch.randn(x.shape, generator=g)
    sn = torch.randn((T,) + tuple(x.shape), generator=g)
    m = full_drmnet(dev, precision, gamma=0.9, epsilon=0.01, max_timesteps=T)
    X, N0, SN = x.to(dev), n0.to(dev), sn.to(dev)
    Lr0, zK, K = m.p_sample_loop(X, [X], [X], verbose=False, noise0=N0, step_noise=SN, early_exit=False)
    assert torch.isfinite(Lr0).all() and rel_l2(Lr0[64].cpu(), Lr0[0].cpu()) > 1e-2  # (distinct rows on both sides of the boundary)
    worst = 0.0
    for r in sorted(set(list(range(0, B, 8)) + [62, 63, 64, 65, 127])):
        xr = X[r:r + 1].contiguous()
        one, _, _ = m.p_sample_loop(xr, [xr], [xr], verbose=False, noise0=N0[r:r + 1].contiguous(), step_noise=SN[:, r:r + 1].contiguous(), early_exit=False)
        worst = max(worst, rel_l2(Lr0[r].cpu(), one[0].cpu()))
    print(f"DRMNet loop B=128, two forked row ranges, {T} steps ({precision}): worst probed row vs the loop on that row alone {worst:.2e}")
    assert worst < SELF_TOL[precision]
    del m
    torch.cuda.empty_cache()


def test_configs2_as_written_bf16_ddim50_graph_replay(dev):
    """BASELINE configs[2]: "DRMNet DDIM 50-step, bf16, hipGraph-captured step" -- ObsNet's DDIM-50 chain (the reference's only DDIM schedule) at
    B = 32 @3x128x256 on bf16 operands: the replayed chain is the eager chain bit for bit, and its final state sits within the bf16 tolerance (3e-2,
    tests/test_gpu_bf16.py) of the same chain in f16x3 (eta = 1, same Philox draws)."""
    import os

    from conftest import GOLD
    from drmnet_amd import ops
    from drmnet_amd.config import instantiate_from_config, load_config
    from drmnet_amd.ddim import DDIMSampler

    root = os.path.dirname(os.path.dirname(GOLD))
    cfg = load_config(os.path.join(root, "configs/obsnet/eval_obsnet.yaml"))["model"]
    cfg["params"].pop("ckpt_path", None)
    cfg["params"].update(use_ema=False)
    obs = instantiate_from_config(cfg)
    synth.load_synth(obs.model.diffusion_model, synth.SEED_OBSNET)
    obs = obs.to(dev)
    B = 32
    cond = synth.synth_refmaps(B, 128, 256, 4242).to(dev) * 2 - 1
    x_T = torch.randn((B, 3, 128, 256), generator=torch.Generator().manual_seed(6)).to(dev)
    smp = DDIMSampler(obs)
    smp.make_schedule(50, ddim_eta=1.0, verbose=False)

    def chain(precision, graph):
        obs.set_precision(precision)
        ops.set_graph_replay(graph)
        try:
            n0 = ops.graph_launches()
            x, _ = smp.ddim_sampling(cond, tuple(x_T.shape), x_T=x_T, seed=9, log_every_t=0, verbose=False)
            return x, ops.graph_launches() - n0
        finally:
            ops.set_graph_replay(False)

    x_eager, n_e = chain("bf16", False)
    x_graph, n_g = chain("bf16", True)
    assert n_e == 0 and n_g == 49  # step 1 eager, then the captured step replayed for steps 2 .. 50
    assert torch.isfinite(x_eager).all() and torch.equal(x_graph, x_eager)
    x_ref, _ = chain("f16x3", False)
    e = rel_l2(x_eager.cpu(), x_ref.cpu())
    rows = ((x_eager - x_ref).flatten(1).norm(dim=1) / x_ref.flatten(1).norm(dim=1)).max().item()
    print(f"configs[2] as written (ObsNet DDIM-50, B=32 @3x128x256, bf16, graph replay == eager): final state vs the f16x3 chain {e:.2e} (worst row {rows:.2e})")
    assert e < MODE_TOL["bf16"] and rows < 2 * MODE_TOL["bf16"]
