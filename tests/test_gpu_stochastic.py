"""GPU parity tests (-m gpu) of the stochastic paths (the rest of the row): stochastic splats
(MGS_SORT_STOCHASTIC; threedgs_raster.frag.slang:265-290, threedgut_raster.frag.slang:150-172), depth of field of the 3DGUT
pipeline (threedgut_raster.frag.slang:104-109, cameras.h.slang:85-108) and temporal accumulation (post.comp.slang), HIP
through the C ABI against the CPU oracle.

The random numbers are a pure function of (pixel, frame_sample_id, splat id, triangle), so a stochastic frame is
deterministic and comparable pixel by pixel: the oracle is fed the splats in the library's storage order so that the
ids that enter the hash are the same.  A pixel's value is ONE splat's colour, so a fragment whose opacity differs in the
last bits between HIP and the oracle (exp vs exp2, fp contraction) can flip a decision and change the whole pixel: the bar
is the fraction of identical pixels, not a PSNR."""
import numpy as np
import pytest

import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth

pytestmark = pytest.mark.gpu
SAME_MIN = 0.995     # fraction of pixels whose colour matches the oracle's within 2e-3
W, H = 640, 400


@pytest.fixture(scope="module")
def stage():
    sc = synth.make_scene(40000, seed=33)
    ss = mgs.SplatSet.from_arrays(**sc)
    scene = mgs.Scene(0)
    scene.add_instance(ss)
    scene.commit()
    n = sc["positions"].shape[0]
    perm = scene.storage_order(0, n)
    sc_p = {k: (v[perm] if v is not None else None) for k, v in sc.items()}
    yield scene, sc, sc_p
    scene.close()


def params(pose, **kw):
    eye = synth.orbit_pose(pose)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H)
    capi.set_camera(p, V, P, eye)
    for k, v in kw.items():
        setattr(p, k, v)
    return p, V, P, eye


def oracle_storage_frame(ob, sc_p, V, P, eye, gut, **fkw):
    """oracle frame with the splats in storage order (ids == the device's ids), drawn in the stable far-to-near order"""
    ps = ob.PreparedSet(sc_p)
    inst = ob.make_instances([(ps, None)])
    ok, oi = ob.key_cull(ob.make_frame(V, P, eye, W, H), inst)
    _, ois = ob.sort_stable(ok, oi)
    fr = ob.make_frame(V, P, eye, W, H, target_fp16=0, **fkw)
    return (ob.render_gut if gut else ob.render)(fr, inst, ois)


@pytest.mark.parametrize("gut", [False, True])
@pytest.mark.parametrize("sample", [0, 7])
def test_stochastic_frame_matches_oracle(stage, ob, gut, sample):
    scene, sc, sc_p = stage
    p, V, P, eye = params(4, sort_mode=capi.SORT_STOCHASTIC, frame_sample_id=sample, target_format=capi.TARGET_RGBA32F,
                          pipeline=capi.PIPELINE_3DGUT if gut else capi.PIPELINE_3DGS)
    out = scene.render(p, want_stats=True)
    img = scene.download_frame(p).astype(np.float32)
    oimg, st = oracle_storage_frame(ob, sc_p, V, P, eye, gut, stochastic=1, frame_sample_id=sample)
    same = np.all(np.abs(img[..., :3] - oimg[..., :3]) <= 2e-3, axis=-1)
    cover = oimg[..., 3] > 0
    print(f"stochastic {'3DGUT' if gut else '3DGS'} sample {sample}: identical pixels {same.mean():.5f}, covered {cover.mean():.3f}, "
          f"alpha agreement {(np.abs(img[..., 3] - oimg[..., 3]) < 0.5).mean():.5f}")
    assert out.error_flags == 0
    assert same.mean() >= SAME_MIN
    # opaque writes: alpha is 1 where a fragment was accepted and 0 elsewhere
    a = img[..., 3]
    assert np.all((a == 0.0) | (a == 1.0)) and (np.abs(a - oimg[..., 3]) < 0.5).mean() >= SAME_MIN
    # a stochastic frame is deterministic, another sample id gives another pattern, and strips equal the full frame
    scene.render(p)
    assert np.array_equal(scene.download_frame(p), img)
    p.frame_sample_id = sample + 1
    scene.render(p)
    other = scene.download_frame(p).astype(np.float32)
    assert (np.abs(other[..., :3] - img[..., :3]).max(axis=-1) > 1e-3).mean() > 0.05
    p.frame_sample_id = sample
    for b, e in ((0, 9), (9, 25)):
        p.strip_row_begin, p.strip_row_end = b, e
        scene.render(p)
        part = scene.download_frame(p)
        assert np.array_equal(part[b * 16:min(e * 16, H)], img[b * 16:min(e * 16, H)]), (b, e)


@pytest.mark.parametrize("gut", [False, True])
def test_stochastic_mean_converges_to_the_blended_frame(stage, ob, gut):
    """E[first accepted fragment] = sum_i alpha_i prod_{j nearer}(1 - alpha_j) c_i: the sorted alpha blend.  The running mean
    (temporal_sampling, post.comp.slang) of K samples must approach the blended frame like 1/sqrt(K)."""
    scene, sc, sc_p = stage
    pipe = capi.PIPELINE_3DGUT if gut else capi.PIPELINE_3DGS
    p, V, P, eye = params(11, target_format=capi.TARGET_RGBA32F, pipeline=pipe)
    scene.render(p)
    blended = scene.download_frame(p).astype(np.float32)
    p.sort_mode = capi.SORT_STOCHASTIC
    p.temporal_sampling = 1
    psnr = {}
    acc = None
    for k in range(64):
        p.frame_sample_id = k
        scene.render(p)
        if k + 1 in (4, 16, 64):
            acc = scene.download_frame(p).astype(np.float32)
            psnr[k + 1] = ob.psnr_rgb(acc, blended)
    print(f"stochastic {'3DGUT' if gut else '3DGS'} running mean vs blended frame: " + ", ".join(f"K={k}: {v:.2f} dB" for k, v in psnr.items()))
    assert psnr[16] > psnr[4] + 4.0 and psnr[64] > psnr[16] + 4.0   # ~6 dB per 4x samples
    assert psnr[64] >= 27.0
    # the accumulated image is the plain mean of the individual samples (fp32 accumulator)
    p.temporal_sampling = 0
    mean = np.zeros_like(blended, dtype=np.float64)
    for k in range(64):
        p.frame_sample_id = k
        scene.render(p)
        mean += scene.download_frame(p).astype(np.float64)
    mean /= 64.0
    assert np.abs(acc - mean).max() <= 2e-5


def test_temporal_accumulation_matches_post_comp(stage, ob):
    """RGBA16F target: the frame handed back is lerp(main, aux1, 1/(id+1)) of post.comp.slang, accumulator in fp32"""
    scene, sc, sc_p = stage
    p, V, P, eye = params(2, sort_mode=capi.SORT_STOCHASTIC)
    singles = []
    for k in range(6):
        p.frame_sample_id = k
        scene.render(p)
        singles.append(scene.download_frame(p).astype(np.float32))
    p.temporal_sampling = 1
    main = np.zeros_like(singles[0])
    for k in range(6):
        p.frame_sample_id = k
        scene.render(p)
        got = scene.download_frame(p).astype(np.float32)
        main = ob.post_accumulate(main, singles[k], k)
        want = main.astype(np.float16).astype(np.float32)
        assert np.abs(got - want).max() <= 1e-3, k   # one fp16 rounding of the output
    # sample 0 restarts the accumulation
    p.frame_sample_id = 0
    scene.render(p)
    assert np.array_equal(scene.download_frame(p).astype(np.float32), singles[0])


@pytest.mark.parametrize("aperture,sample", [(0.02, 0), (0.02, 5), (0.0, 3)])
def test_gut_depth_of_field_matches_oracle(stage, ob, aperture, sample):
    scene, sc, sc_p = stage
    focus = 3.0
    p, V, P, eye = params(6, pipeline=capi.PIPELINE_3DGUT, dof_mode=capi.DOF_FIXED_FOCUS, focus_dist=focus, aperture=aperture,
                          frame_sample_id=sample)
    out = scene.render(p, want_stats=True)
    img = scene.download_frame(p).astype(np.float32)
    ps = ob.PreparedSet(sc_p)
    inst = ob.make_instances([(ps, None)])
    ok, oi = ob.key_cull(ob.make_frame(V, P, eye, W, H), inst)
    _, ois = ob.sort_stable(ok, oi)
    fr = ob.make_frame(V, P, eye, W, H, target_fp16=1, dof_mode=1, focus_dist=focus, aperture=aperture, frame_sample_id=sample)
    oimg, st = ob.render_gut(fr, inst, ois)
    psnr = ob.psnr_rgb(img, oimg)
    err = np.abs(img[..., :3] - oimg[..., :3])
    p.dof_mode = capi.DOF_DISABLED
    scene.render(p)
    sharp = scene.download_frame(p).astype(np.float32)
    d = ob.psnr_rgb(img, sharp)
    print(f"3DGUT depth of field aperture {aperture} sample {sample}: PSNR vs oracle {psnr:.2f} dB, max abs {err.max():.4f}; "
          f"vs the pinhole frame {d:.2f} dB")
    assert out.error_flags == 0 and psnr >= 50.0 and err.max() <= 3e-2
    if aperture == 0.0:
        assert d >= 60.0      # a closed aperture is the pinhole camera (up to the renormalised direction)
    else:
        assert d < 45.0       # the lens visibly perturbs the rays


def test_dof_needs_the_gut_pipeline_and_bad_modes_are_rejected(stage):
    scene, sc, sc_p = stage
    p, *_ = params(0, dof_mode=capi.DOF_FIXED_FOCUS)
    with pytest.raises(mgs.MgsError):
        scene.render(p)
    p, *_ = params(0, sort_mode=2)
    with pytest.raises(mgs.MgsError):
        scene.render(p)
    p, *_ = params(0, frame_sample_id=-1)
    with pytest.raises(mgs.MgsError):
        scene.render(p)


@pytest.mark.parametrize("degree", [0, 1, 3, 4, 5, 8])
def test_gut_kernel_degrees_match_oracle(stage, ob, degree):
    """KERNEL_DEGREE (shaderio.h:112-119): the generalised Gaussians of particleRayMaxKernelResponse in the 3DGUT compositor"""
    scene, sc, sc_p = stage
    p, V, P, eye = params(8, pipeline=capi.PIPELINE_3DGUT, kernel_degree=degree)
    out = scene.render(p, want_stats=True)
    img = scene.download_frame(p).astype(np.float32)
    oimg, st = oracle_storage_frame(ob, sc_p, V, P, eye, True, kernel_degree=degree)
    oimg = oimg.astype(np.float16).astype(np.float32)
    psnr = ob.psnr_rgb(img, oimg)
    err = np.abs(img[..., :3] - oimg[..., :3])
    print(f"3DGUT kernel degree {degree}: PSNR {psnr:.2f} dB, max abs {err.max():.4f}")
    assert out.error_flags == 0 and psnr >= 50.0 and err.max() <= 3e-2
    p.kernel_degree = 7
    with pytest.raises(mgs.MgsError):
        scene.render(p)
