"""Image IO at the edges of the pipeline (reference: utils/file_io.py:10-56, which goes through OpenCV).

OpenCV is not part of this stack, so the two formats the reference's sample inputs use are decoded here:

* OpenEXR (`load_exr`, `save_exr`): single-part scanline files, NO / ZIPS / ZIP compression, HALF / FLOAT / UINT channels
  (data/sample/image.exr is ZIP, FLOAT, channels B G R, 256x256).  ~80 lines of `zlib` + the EXR byte predictor and
  even/odd interleave.  Returned as RGB float32 [H, W, 3] like the reference's `cv2.imread(..., -1)[..., :3]` + BGR2RGB.
* PNG (`load_png`, `save_png`) through Pillow, with the reference's conventions (values / 255, optional alpha mask).

Parity: no OpenEXR implementation exists in this image to compare with ("parity unpinned" for the reader itself); the
tests pin it by a write -> read round trip, by the file's own invariants (window size, channel list) and by the
downstream `refmap_mask_make` golden that was generated from the decoded sample.
"""
from __future__ import annotations

import struct
import zlib
from pathlib import Path
from typing import Dict, Tuple, Union

import numpy as np
import torch

_PIX_DTYPE = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}
_COMPRESSION_LINES = {0: 1, 2: 1, 3: 16}  # NO, ZIPS, ZIP


def _read_header(b: bytes) -> Tuple[Dict[str, Tuple[str, bytes]], int]:
    magic, version = struct.unpack_from("<II", b, 0)
    if magic != 20000630:
        raise ValueError("not an OpenEXR file")
    if version & 0x1E00:  # tiled / long names / deep / multipart bits 9..12
        if version & 0x200 or version & 0x800 or version & 0x1000:
            raise ValueError("only single-part scanline OpenEXR files are supported")
    p = 8
    attrs: Dict[str, Tuple[str, bytes]] = {}
    while True:
        e = b.index(b"\0", p)
        name = b[p:e].decode()
        p = e + 1
        if not name:
            break
        e = b.index(b"\0", p)
        typ = b[p:e].decode()
        p = e + 1
        (size,) = struct.unpack_from("<i", b, p)
        p += 4
        attrs[name] = (typ, b[p : p + size])
        p += size
    return attrs, p


def _undo_zip(raw: bytes) -> bytes:
    """OpenEXR ZIP post-processing: byte-delta predictor, then the two halves are the even and the odd bytes."""
    t = np.frombuffer(raw, dtype=np.uint8).astype(np.int64)
    t[1:] -= 128
    t = (np.cumsum(t) & 0xFF).astype(np.uint8)
    n = t.size
    out = np.empty(n, dtype=np.uint8)
    half = (n + 1) // 2
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def read_exr_channels(path: Union[str, Path]) -> Dict[str, np.ndarray]:
    b = Path(path).read_bytes()
    attrs, p = _read_header(b)
    comp = attrs["compression"][1][0]
    if comp not in _COMPRESSION_LINES:
        raise ValueError(f"OpenEXR compression {comp} is not supported (NO, ZIPS and ZIP are)")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    chans = []
    cl = attrs["channels"][1]
    q = 0
    while cl[q] != 0:
        e = cl.index(b"\0", q)
        name = cl[q:e].decode()
        q = e + 1
        ptype, _plinear, xs, ys = struct.unpack_from("<iB3xii", cl, q)
        q += 16
        if xs != 1 or ys != 1:
            raise ValueError("sub-sampled OpenEXR channels are not supported")
        chans.append((name, _PIX_DTYPE[ptype]))
    lines_per_block = _COMPRESSION_LINES[comp]
    n_blocks = (H + lines_per_block - 1) // lines_per_block
    offsets = struct.unpack_from(f"<{n_blocks}Q", b, p)
    out = {name: np.zeros((H, W), dtype=dt) for name, dt in chans}
    row_bytes = sum(dt.itemsize for _, dt in chans) * W
    for off in offsets:
        y, size = struct.unpack_from("<ii", b, off)
        data = b[off + 8 : off + 8 + size]
        rows = min(lines_per_block, y1 - y + 1)
        want = rows * row_bytes
        if comp != 0 and size < want:
            data = _undo_zip(zlib.decompress(data))
        if len(data) != want:
            raise ValueError("corrupt OpenEXR chunk")
        q = 0
        for ry in range(rows):
            for name, dt in chans:
                out[name][y - y0 + ry] = np.frombuffer(data, dtype=dt, count=W, offset=q)
                q += W * dt.itemsize
    return out


def load_exr(path: Union[str, Path], as_torch: bool = False, channel_first: bool = False):
    """utils/file_io.py:10-17 (alpha ignored; RGB float32)."""
    ch = read_exr_channels(path)
    if all(k in ch for k in "RGB"):
        img = np.stack([ch["R"], ch["G"], ch["B"]], -1)
    else:  # single-channel files come back grey, as OpenCV does
        (only,) = list(ch.values())[:1]
        img = np.stack([only] * 3, -1)
    img = np.ascontiguousarray(img.astype(np.float32))
    if channel_first:
        img = img.transpose(2, 0, 1)
    return torch.from_numpy(np.ascontiguousarray(img)) if as_torch else img


def save_exr(path: Union[str, Path], img, channel_first: bool = False, compress: bool = True) -> None:
    """utils/file_io.py:20-25: RGB float32 [H, W, 3] -> scanline OpenEXR (ZIP or uncompressed, FLOAT, channels B G R)."""
    if isinstance(img, torch.Tensor):
        img = img.detach().cpu().numpy()
    if channel_first:
        img = img.transpose(1, 2, 0)
    img = np.asarray(img, dtype="<f4")
    H, W = img.shape[:2]
    names = ["B", "G", "R"]
    planes = {"R": img[..., 0], "G": img[..., 1], "B": img[..., 2]}

    def attr(name: str, typ: str, val: bytes) -> bytes:
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val

    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)
    head = struct.pack("<II", 20000630, 2)
    head += attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([3 if compress else 0]))
    head += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0")
    head += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0))
    head += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = 16 if compress else 1
    chunks = []
    for y in range(0, H, lines):
        rows = min(lines, H - y)
        raw = b"".join(planes[n][y + r].tobytes() for r in range(rows) for n in names)
        data = raw
        if compress:
            t = np.frombuffer(raw, dtype=np.uint8)
            t = np.concatenate([t[0::2], t[1::2]]).astype(np.int64)
            d = t.copy()
            d[1:] = (t[1:] - t[:-1] + 128 + 256) & 0xFF
            z = zlib.compress(d.astype(np.uint8).tobytes())
            if len(z) < len(raw):
                data = z
        chunks.append(struct.pack("<ii", y, len(data)) + data)
    table_at = len(head)
    pos = table_at + 8 * len(chunks)
    offs = []
    for c in chunks:
        offs.append(pos)
        pos += len(c)
    Path(path).write_bytes(head + struct.pack(f"<{len(offs)}Q", *offs) + b"".join(chunks))


def load_png(path: Union[str, Path], as_torch: bool = False, channel_first: bool = False):
    """utils/file_io.py:46-56: 8/16-bit PNG -> float64 in [0, 1] (value / 255 exactly as the reference does)."""
    from PIL import Image

    im = Image.open(str(path))
    if im.mode == "P":
        im = im.convert("RGBA" if "transparency" in im.info else "RGB")
    img = np.array(im)
    if img.dtype == np.bool_:
        img = img.astype(np.uint8) * 255
    if channel_first:
        if img.ndim == 2:
            img = img[:, :, None]
        img = img.transpose(2, 0, 1)
    if as_torch:
        return torch.from_numpy(np.ascontiguousarray(img)) / 255.0
    return img / 255.0


def save_png(path: Union[str, Path], ldr, channel_first: bool = False, mask=None) -> None:
    """utils/file_io.py:28-43: RGB in [0, 1] (+ optional [H, W] alpha mask) -> 8-bit PNG (values * 255, truncated as OpenCV's
    saturate_cast rounds: to nearest)."""
    from PIL import Image

    if isinstance(ldr, torch.Tensor):
        ldr = ldr.detach().cpu().numpy()
    if isinstance(mask, torch.Tensor):
        mask = mask.detach().cpu().numpy()
    if channel_first:
        ldr = ldr.transpose(1, 2, 0)
    ldr = np.asarray(ldr)[:, :, :3]
    if mask is not None:
        mask = np.asarray(mask, dtype=ldr.dtype)
        if mask.ndim == 2:
            mask = mask[:, :, None]
        ldr = np.concatenate([ldr, mask], axis=-1)
    u8 = np.clip(np.rint(ldr * 255.0), 0, 255).astype(np.uint8)
    Image.fromarray(u8, "RGBA" if u8.shape[-1] == 4 else "RGB").save(str(path))
