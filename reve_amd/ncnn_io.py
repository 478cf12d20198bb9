"""ncnn .param / .bin writer + reader for SRVGGNetCompact models.

reve loads `realesr-animevideov3-x{s}.param/.bin` only indirectly, by naming the model on
the `realesrgan-ncnn-vulkan` command line (reve-shared/src/lib.rs:140-141,
reve-gui/src-tauri/src/commands.rs:58-61).  The real files are not in the reference
(reve-gui/.gitignore:27-30); this module writes byte-compatible files from synthetic
weights so that the library's C++ loader (csrc/model.cpp) is exercised on the same format,
and reads them back (used by tests to cross-check the C++ parser).

Format [UPSTREAM-RECALL, Tencent/ncnn]:
  .param  text: "7767517", "<layers> <blobs>", then per layer
          "Type name n_in n_out in_blobs... out_blobs... key=value..."
  .bin    per Convolution: u32 tag (0x01306B47 -> fp16 payload padded to 4 B; 0 -> fp32),
          weights OIHW, then fp32 bias; per PReLU: fp32 slopes (no tag).
"""
from __future__ import annotations

import io
import os
import struct
import numpy as np

FP16_TAG = 0x01306B47
MAGIC = "7767517"


def build_param_text(scale: int, n_body: int = 16, feat: int = 64) -> str:
    lines = []
    blobs = 0

    def add(s):
        lines.append(s)

    add("Input            data                     0 1 data")
    add("Split            splitncnn_input0         1 2 data data_splitncnn_0 data_splitncnn_1")
    blobs += 3
    prev = "data_splitncnn_1"
    idx = 0
    cin = 3
    for l in range(n_body + 2):
        cout = feat if l <= n_body else 3 * scale * scale
        name = f"Conv_{idx}"
        out = f"conv{idx}"
        add(f"Convolution      {name:<24} 1 1 {prev} {out} 0={cout} 1=3 11=3 2=1 12=1 3=1 13=1 "
            f"4=1 14=1 15=1 16=1 5=1 6={cout * cin * 9}")
        blobs += 1
        prev = out
        idx += 1
        if l <= n_body:
            name = f"PRelu_{idx}"
            out = f"prelu{idx}"
            add(f"PReLU            {name:<24} 1 1 {prev} {out} 0={feat}")
            blobs += 1
            prev = out
            idx += 1
        cin = feat
    add(f"PixelShuffle     DepthToSpace_{idx:<11} 1 1 {prev} shuffle 0={scale}")
    blobs += 1
    add(f"Interp           Resize_{idx + 1:<17} 1 1 data_splitncnn_0 nearest 0=1 1={float(scale):e} 2={float(scale):e} 3=0 4=0 6=0")
    blobs += 1
    add(f"BinaryOp         Add_{idx + 2:<20} 2 1 shuffle nearest output 0=0")
    blobs += 1
    return MAGIC + "\n" + f"{len(lines)} {blobs}\n" + "\n".join(lines) + "\n"


def _conv_blob(w: np.ndarray, b: np.ndarray, fp16: bool) -> bytes:
    out = io.BytesIO()
    flat = np.ascontiguousarray(w, dtype=np.float32).reshape(-1)
    if fp16:
        out.write(struct.pack("<I", FP16_TAG))
        payload = flat.astype(np.float16).tobytes()
        out.write(payload)
        out.write(b"\0" * ((-len(payload)) % 4))
    else:
        out.write(struct.pack("<I", 0))
        out.write(flat.tobytes())
    out.write(np.ascontiguousarray(b, dtype=np.float32).tobytes())
    return out.getvalue()


def build_bin(w: dict, fp16: bool = True) -> bytes:
    out = io.BytesIO()
    out.write(_conv_blob(w["w_first"], w["b_first"], fp16))
    out.write(np.ascontiguousarray(w["a_first"], dtype=np.float32).tobytes())
    for l in range(w["n_body"]):
        out.write(_conv_blob(w["w_body"][l], w["b_body"][l], fp16))
        out.write(np.ascontiguousarray(w["a_body"][l], dtype=np.float32).tobytes())
    out.write(_conv_blob(w["w_last"], w["b_last"], fp16))
    return out.getvalue()


def write_model(model_dir: str, name: str, w: dict, fp16: bool = True) -> tuple[str, str]:
    """Writes <model_dir>/<name>.param and .bin; returns the two paths."""
    os.makedirs(model_dir, exist_ok=True)
    p = os.path.join(model_dir, name + ".param")
    b = os.path.join(model_dir, name + ".bin")
    with open(p, "w") as f:
        f.write(build_param_text(w["scale"], w["n_body"]))
    with open(b, "wb") as f:
        f.write(build_bin(w, fp16))
    return p, b


def read_model_files(model_dir: str, name: str) -> tuple[bytes, bytes]:
    """The raw bytes of <model_dir>/<name>.param and .bin (what reve_config.param_data / bin_data take)."""
    with open(os.path.join(model_dir, name + ".param"), "rb") as f:
        p = f.read()
    with open(os.path.join(model_dir, name + ".bin"), "rb") as f:
        b = f.read()
    return p, b


def parse_model(param_text: str, bin_bytes: bytes) -> dict:
    """Reads an SRVGGNetCompact ncnn model back into the dict layout of synth.make_weights."""
    toks = param_text.split("\n")
    if toks[0].strip() != MAGIC:
        raise ValueError("bad ncnn magic")
    convs, prelus, scale = [], [], None
    for line in toks[2:]:
        f = line.split()
        if not f:
            continue
        kv = dict(t.split("=", 1) for t in f if "=" in t and not t.startswith("="))
        if f[0] == "Convolution":
            convs.append((int(kv["0"]), int(kv["6"])))
        elif f[0] == "PReLU":
            prelus.append(int(kv["0"]))
        elif f[0] == "PixelShuffle":
            scale = int(kv["0"])
    if scale is None or len(convs) < 3 or len(prelus) != len(convs) - 1:
        raise ValueError("not an SRVGGNetCompact graph")
    off = 0

    def conv(co, n):
        nonlocal off
        (tag,) = struct.unpack_from("<I", bin_bytes, off)
        off += 4
        if tag == FP16_TAG:
            a = np.frombuffer(bin_bytes, dtype=np.float16, count=n, offset=off).astype(np.float32)
            off += (n * 2 + 3) // 4 * 4
        elif tag == 0:
            a = np.frombuffer(bin_bytes, dtype=np.float32, count=n, offset=off).copy()
            off += n * 4
        else:
            raise ValueError(f"unsupported weight tag {tag:#x}")
        b = np.frombuffer(bin_bytes, dtype=np.float32, count=co, offset=off).copy()
        off += co * 4
        return a, b

    def vec(n):
        nonlocal off
        a = np.frombuffer(bin_bytes, dtype=np.float32, count=n, offset=off).copy()
        off += n * 4
        return a

    feat = convs[0][0]
    n_body = len(convs) - 2
    w = {"scale": scale, "n_body": n_body}
    a, b = conv(*convs[0])
    w["w_first"], w["b_first"] = a.reshape(feat, 3, 3, 3), b
    w["a_first"] = vec(prelus[0])
    wb, bb, ab = [], [], []
    for l in range(n_body):
        a, b = conv(*convs[1 + l])
        wb.append(a.reshape(feat, feat, 3, 3))
        bb.append(b)
        ab.append(vec(prelus[1 + l]))
    w["w_body"], w["b_body"], w["a_body"] = np.stack(wb), np.stack(bb), np.stack(ab)
    a, b = conv(*convs[-1])
    w["w_last"], w["b_last"] = a.reshape(convs[-1][0], feat, 3, 3), b
    if off != len(bin_bytes):
        raise ValueError(f"trailing bytes in .bin: parsed {off} of {len(bin_bytes)}")
    return w
