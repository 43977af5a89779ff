"""Edge cases of the drop-in boundary on the GPU: empty and single-row batches, shapes the engine cannot tile, wrong channel
counts -- what the reference does with them (empty result / exception), never a crash and never a silent fallback."""
import ctypes as C

import numpy as np
import pytest
import torch

from drmnet_amd import _lib, synth
from drmnet_amd.unet import EncoderUNetModel, UNetModel

pytestmark = pytest.mark.gpu

TINY = dict(image_size=16, in_channels=6, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=[2], channel_mult=(1, 2),
            conv_resample=False, num_heads=1)


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _net(cls, dev, **over):
    torch.manual_seed(0)
    m = cls(**{**TINY, **over})
    m.load_state_dict(synth.synth_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()], 7))
    return m.to(dev)


def test_empty_batch_is_an_empty_result(dev):
    net = _net(UNetModel, dev)
    x = torch.zeros((0, 6, 16, 16), device=dev)
    out = net(x, timesteps=torch.zeros((0,), dtype=torch.int64, device=dev))
    assert tuple(out.shape) == (0, 3, 16, 16)
    out = net.forward_parts(x[:, :3], x[:, 3:], timesteps=torch.zeros((0,), dtype=torch.int64, device=dev))
    assert tuple(out.shape) == (0, 3, 16, 16)
    enc = _net(EncoderUNetModel, dev, out_channels=5)
    assert tuple(enc(x, torch.zeros((0,), dtype=torch.int64, device=dev)).shape) == (0, 5)
    # row gather that selects nothing (every row converged): same
    rows = torch.zeros((0,), dtype=torch.int32, device=dev)
    full = torch.randn((3, 6, 16, 16), device=dev)
    assert tuple(net.forward_parts(full[:, :3], full[:, 3:], timesteps=torch.zeros((0,), dtype=torch.int64, device=dev), rows=rows).shape) == (0, 3, 16, 16)


def test_single_row_and_ragged_batches_match_the_batched_rows(dev):
    """Batch sizes that do not fill a multi-image tile (1, 3, 5 rows on 16x16 / 8x8 maps) give the rows of the full batch."""
    net = _net(UNetModel, dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn((8, 6, 16, 16), generator=g).to(dev)
    t = torch.randint(0, 1000, (8,), generator=g).to(dev)
    ref = net(x, timesteps=t)
    for n in (1, 3, 5):
        got = net(x[:n].contiguous(), timesteps=t[:n].contiguous())
        assert torch.allclose(got, ref[:n], rtol=0, atol=2e-5 * float(ref.abs().max())), n


def test_untileable_map_and_wrong_channels_raise(dev):
    net = _net(UNetModel, dev)
    t = torch.zeros((1,), dtype=torch.int64, device=dev)
    with pytest.raises(RuntimeError):  # 7x6 maps: the Downsample would see an odd height (the reference fails there too: skip shapes differ)
        net(torch.zeros((1, 6, 7, 6), device=dev), timesteps=t)
    with pytest.raises(RuntimeError):
        net(torch.zeros((1, 5, 16, 16), device=dev), timesteps=t)
    with pytest.raises(ValueError):
        net(torch.zeros((1, 6, 16, 16), device=dev))  # neither timesteps nor t_emb (openaimodel.py:741-743)
    with pytest.raises(RuntimeError):  # host tensors never reach a CPU path
        net(torch.zeros((1, 6, 16, 16)), timesteps=t.cpu())
    # the engine is still usable after the errors
    out = net(torch.zeros((1, 6, 16, 16), device=dev), timesteps=t)
    assert torch.isfinite(out).all()


def test_c_abi_rejects_bad_arguments_with_a_message(dev):
    L = _lib.lib()
    net = _net(UNetModel, dev)
    net(torch.zeros((1, 6, 16, 16), device=dev), timesteps=torch.zeros((1,), dtype=torch.int64, device=dev))  # weights loaded
    x = torch.zeros((1, 6, 16, 16), device=dev)
    out = torch.empty((1, 3, 16, 16), device=dev)
    ti = torch.zeros((1,), dtype=torch.int64, device=dev)
    ws = torch.empty(1024, dtype=torch.uint8, device=dev)  # far too small
    rc = L.drm_unet_forward(net._h, x.data_ptr(), 6, None, 0, None, None, ti.data_ptr(), None, out.data_ptr(), 1, 16, 16, ws.data_ptr(), ws.numel(),
                            _lib.stream_ptr(dev))
    assert rc != 0 and len(L.drm_last_error()) > 0
    rc = L.drm_unet_forward(net._h, x.data_ptr(), 6, None, 0, None, None, ti.data_ptr(), None, out.data_ptr(), 0, 16, 16, ws.data_ptr(), ws.numel(),
                            _lib.stream_ptr(dev))
    assert rc != 0 and len(L.drm_last_error()) > 0  # N = 0 is the caller's business (the host mirror returns early)
    torch.cuda.synchronize()


def test_two_handles_on_two_streams_from_two_threads(dev):
    """include/drmnet_hip.h: entry points take the stream to launch on and are re-entrant across distinct handles and streams (error strings
    are thread-local, per-kernel attributes are set idempotently, a handle owns its weights and the caller its workspace).  Two networks
    driven from two host threads on two non-default streams, interleaved, must give what each gives alone."""
    import threading

    nets = [_net(UNetModel, dev).set_precision("f16x3"), _net(UNetModel, dev, model_channels=64).set_precision("fp32")]
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn((n, 6, 16, 32), generator=g).to(dev) for n in (3, 5)]
    ts = [torch.randint(0, 1000, (n,), generator=g).to(dev) for n in (3, 5)]
    refs = [m(x, timesteps=t).clone() for m, x, t in zip(nets, xs, ts)]
    torch.cuda.synchronize()
    outs, errs = [[], []], []

    def worker(k):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(25):
                    outs[k].append(nets[k](xs[k], timesteps=ts[k]))
            st.synchronize()
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errs.append(e)

    th = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert not errs, errs
    for k in (0, 1):
        assert len(outs[k]) == 25
        for o in outs[k]:
            assert torch.allclose(o, refs[k], rtol=0, atol=2e-6 * float(refs[k].abs().max()))
