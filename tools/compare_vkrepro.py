"""Closes the loop of tests/golden/vkrepro on a Vulkan box: PSNR of a screenshot of the REFERENCE against the expected frame of a case.

    python tools/compare_vkrepro.py <case | path/to/expected.npy> <reference screenshot: .hdr | .npy | .png> [--flip-y]

  <case>       base | base_3dgut | u8_storage | fisheye150_3dgut | msaa_3dgs | two_instances_trs  (tests/golden/vkrepro/...;
               camera.json["reference_command"] of the case says how the reference produces the screenshot)
  screenshot   what GaussianSplattingUI::saveVisualizationImageToFile writes (src/gaussian_splatting_ui.cpp:508-540): ".hdr" = the
               colour target read back as RGBA32F and stored by stb as Radiance RGBE (8-bit mantissas with a shared exponent: about
               2^-9 relative per pixel, which caps the PSNR near 55 dB — use it, the 8-bit formats cap it near 48 dB); ".npy" = a
               float array [H][W][3|4] dumped by other means.
PSNR as shaders/image_compare_metric.comp.slang:116-130 defines it: MSE = sum over pixels of |rgb_ref - rgb_cur|^2 / (W * H * 3),
PSNR = 10 log10(1 / MSE) (peak 1.0), alpha ignored.  north_star bar: >= 40 dB.  Ties between equal depth keys are drawn in
nondeterministic order by the reference (dist.comp.slang:137-139) and its rasteriser snaps vertices to 1/256 px: a few tenths of a
dB of run-to-run noise are expected.  Pure numpy: nothing of the product or the oracle is imported.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"base": ("", "expected_rgba16f.npy"), "base_3dgut": ("", "expected_3dgut_rgba16f.npy")}
for _c in ("u8_storage", "fisheye150_3dgut", "msaa_3dgs", "two_instances_trs"):
    CASES[_c] = (_c, "expected_rgba16f.npy")


def read_hdr(path):
    """Radiance RGBE (.hdr / .pic) -> float32 [H][W][3]; flat and new-style run-length encoded scanlines (what stb_image_write emits)"""
    with open(path, "rb") as f:
        data = f.read()
    pos, fmt_ok = 0, False
    while True:
        end = data.index(b"\n", pos)
        line = data[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if line.startswith("FORMAT=") and "32-bit_rle_rgbe" in line:
            fmt_ok = True
        if line == "":
            break
    end = data.index(b"\n", pos)
    res = data[pos:end].decode("ascii").split()
    pos = end + 1
    if len(res) != 4 or res[0] not in ("-Y", "+Y") or res[2] != "+X":
        raise ValueError(f"{path}: unsupported resolution line {res}")
    H, W = int(res[1]), int(res[3])
    if not fmt_ok:
        print(f"warning: {path} does not declare FORMAT=32-bit_rle_rgbe", file=sys.stderr)
    buf = np.frombuffer(data, np.uint8, offset=pos)
    rgbe = np.zeros((H, W, 4), np.uint8)
    i = 0
    for y in range(H):
        if W >= 8 and W < 32768 and buf[i] == 2 and buf[i + 1] == 2 and (int(buf[i + 2]) << 8 | int(buf[i + 3])) == W:
            i += 4
            for ch in range(4):
                x = 0
                while x < W:
                    n = int(buf[i])
                    i += 1
                    if n > 128:  # run
                        n -= 128
                        rgbe[y, x:x + n, ch] = buf[i]
                        i += 1
                    else:
                        rgbe[y, x:x + n, ch] = buf[i:i + n]
                        i += n
                    x += n
        else:
            rgbe[y] = buf[i:i + 4 * W].reshape(W, 4)
            i += 4 * W
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0).astype(np.float32)  # 2^(e - 128 - 8)
    img = rgbe[..., :3].astype(np.float32) * scale[..., None]
    return img[::-1].copy() if res[0] == "+Y" else img


def write_hdr(path, img):
    """flat (unencoded) RGBE writer: used by the self-test of this tool (tests/test_host_cpu.py)"""
    img = np.asarray(img, np.float32)[..., :3]
    H, W = img.shape[:2]
    m = img.max(axis=2)
    mant, ex = np.frexp(np.where(m > 1e-32, m, 1.0))
    sc = np.where(m > 1e-32, mant * 256.0 / np.where(m > 1e-32, m, 1.0), 0.0)
    rgbe = np.zeros((H, W, 4), np.uint8)
    rgbe[..., :3] = np.clip(img * sc[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(m > 1e-32, ex + 128, 0).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {H} +X {W}\n".encode())
        f.write(rgbe.tobytes())


def load_image(path):
    ext = os.path.splitext(path)[1].lower()
    if ext in (".hdr", ".pic"):
        return read_hdr(path)
    if ext == ".npy":
        return np.load(path).astype(np.float32)[..., :3]
    if ext in (".png", ".bmp", ".jpg", ".jpeg"):
        from PIL import Image  # 8-bit screenshots: sRGB-free UNORM copies of the colour target
        return np.asarray(Image.open(path).convert("RGB"), np.float32) / 255.0
    raise ValueError(f"unsupported image type {ext}")


def psnr_rgb(ref, cur):
    """image_compare_metric.comp.slang:116-130 (+ its host side: PSNR = 10 log10(1 / MSE), 99.99 for identical images)"""
    d = ref[..., :3].astype(np.float64) - cur[..., :3].astype(np.float64)
    mse = float((d * d).sum() / (ref.shape[0] * ref.shape[1] * 3))
    return 99.99 if mse <= 0.0 else min(99.99, 10.0 * np.log10(1.0 / mse)), mse


def main(argv):
    if len(argv) < 3:
        print(__doc__)
        return 2
    what, shot = argv[1], argv[2]
    if what in CASES:
        sub, name = CASES[what]
        exp_path = os.path.join(ROOT, "tests", "golden", "vkrepro", sub, name)
    else:
        exp_path = what
    want = np.load(exp_path).astype(np.float32)
    got = load_image(shot)
    if os.path.splitext(shot)[1].lower() in (".hdr", ".pic", ".png", ".bmp", ".jpg", ".jpeg"):
        want = np.maximum(want, 0.0)  # these formats cannot hold the (slightly) negative colours an SH sum may produce
    if "--flip-y" in argv:
        got = got[::-1]
    if got.shape[:2] != want.shape[:2]:
        print(f"size mismatch: expected {want.shape[1]}x{want.shape[0]}, screenshot {got.shape[1]}x{got.shape[0]} "
              f"(set the viewport to the size in camera.json)")
        return 1
    psnr, mse = psnr_rgb(want, got)
    err = np.abs(want[..., :3] - got[..., :3])
    flipped, _ = psnr_rgb(want, got[::-1])
    print(f"{os.path.relpath(exp_path, ROOT)} vs {shot}: PSNR {psnr:.2f} dB (MSE {mse:.3e}), max abs {err.max():.4f}, "
          f"99.9th pct {np.percentile(err, 99.9):.4f}; north_star bar 40 dB -> {'PASS' if psnr >= 40.0 else 'FAIL'}")
    if flipped > psnr + 3.0:
        print(f"(the vertically flipped screenshot scores {flipped:.2f} dB: rerun with --flip-y)")
    return 0 if psnr >= 40.0 else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv))
