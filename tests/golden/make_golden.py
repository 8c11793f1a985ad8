"""Generates the committed golden vectors under tests/golden/ (run once, here, on CPU):

    python tests/golden/make_golden.py          (everything)
    python tests/golden/make_golden.py traj     (only reg3d_traj.npz, from the committed reg3d.npz)

Sources of truth
  * powell_ref.npz      -- produced by oracle/_ref/libpowell_ref.so, i.e. the reference's own
                           src/api_powell.c compiled unmodified (needs /root/reference at build time).
  * everything else     -- produced by the NumPy restatement in oracle/ (the reference ships no
                           tests or vectors for these paths, and its decon / registration sources
                           cannot be built in this image; see DESIGN.md "Oracle").
A fixture is data only: seeded inputs and the expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import decon_oracle as do  # noqa: E402
from oracle import powell_oracle as po  # noqa: E402
from oracle import reg_oracle as ro  # noqa: E402
from tests.powell_cases import make_cost, start_point  # noqa: E402

f32 = np.float32


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def decon_cases():
    # (1) small PSF, non-FFT-friendly size 50x36x20 -> box 64x64x32 (pad / crop exercised)
    shape = (20, 36, 50)
    psf_a = do.gaussian_psf((9, 9, 9), (2.0, 1.2, 1.2))
    psf_b = do.gaussian_psf((9, 9, 9), (1.2, 1.2, 2.0))
    img_a = do.synthetic_view(shape, psf_a, seed=0, noise_seed=1, n_blobs=12)
    img_b = do.synthetic_view(shape, psf_b, seed=0, noise_seed=2, n_blobs=12)
    save("decon_sv_small.npz", img=img_a, psf=psf_a, iters=np.int32(5),
         expected=do.decon_singleview(img_a, psf_a, 5),
         expected_const=do.decon_singleview(img_a, psf_a, 5, const_init=True))
    save("decon_dv_small.npz", img1=img_a, img2=img_b, psf1=psf_a, psf2=psf_b, iters=np.int32(4),
         expected=do.decon_dualview(img_a, img_b, psf_a, psf_b, 4),
         expected_const=do.decon_dualview(img_a, img_b, psf_a, psf_b, 4, const_init=True))
    # (2) PSF box == FFT box (16x32x32), FFT-friendly image: no pad, no crop
    shape2 = (16, 32, 32)
    psf_full = do.gaussian_psf(shape2, (1.5, 1.5, 1.5))
    img2 = do.synthetic_view(shape2, psf_full, seed=3, noise_seed=4, n_blobs=8)
    save("decon_sv_psf_eq_fft.npz", img=img2, psf=psf_full, iters=np.int32(10),
         expected=do.decon_singleview(img2, psf_full, 10))
    # (3) unmatched back projector + even-sized PSF (origin = s/2)
    psf_even = do.gaussian_psf((8, 10, 12), (1.5, 1.5, 2.0))
    bp = do.gaussian_psf((8, 10, 12), (1.0, 1.0, 1.2))
    save("decon_sv_unmatched.npz", img=img_a, psf=psf_even, psf_bp=bp, iters=np.int32(3),
         expected=do.decon_singleview(img_a, psf_even, 3, unmatch=True, psf_bp=bp))
    # (4) PSF larger than the FFT box on one axis (genOTFgpu's alignsize branch)
    psf_big = do.gaussian_psf((40, 9, 9), (3.0, 1.2, 1.2))
    save("decon_sv_bigpsf.npz", img=img_a, psf=psf_big, iters=np.int32(2),
         expected=do.decon_singleview(img_a, psf_big, 2))
    # OTF alone (both parities of PSF extent)
    for tag, psf in (("odd", psf_a), ("even", psf_even)):
        F = do.fft_shape(shape)
        otf = do.gen_otf(psf, F)
        otf_bp = do.gen_otf(do.flip3(psf), F)
        save(f"otf_{tag}.npz", psf=psf, box=np.array(F, dtype=np.int32), otf=otf, otf_flipped=otf_bp)


def geometry_cases():
    rng = np.random.default_rng(7)
    a = rng.standard_normal((7, 10, 13)).astype(f32)
    save("geometry.npz", a=a,
         pad_stack=do.pad_stack(a, (16, 16, 16)),
         crop=do.crop_stack(do.pad_stack(a, (16, 16, 16)), (7, 10, 13)),
         align_grow=do.align_size(a, (9, 12, 16)),
         align_shrink=do.align_size(a, (4, 7, 8)),
         align_mixed=do.align_size(a, (12, 5, 13)),
         pad_psf=do.pad_psf(a, (16, 16, 16)),
         flip=do.flip3(a))


def affine_cases():
    rng = np.random.default_rng(11)
    src = do.blob_volume((24, 40, 48), n_blobs=15, seed=5) + rng.standard_normal((24, 40, 48)).astype(f32)
    tgt = do.blob_volume((24, 40, 48), n_blobs=15, seed=5)
    mats = []
    ident = np.zeros(12, dtype=f32)
    ident[0] = ident[5] = ident[10] = 1
    mats.append(ident)
    m = ident.copy(); m[3], m[7], m[11] = 2.25, -1.5, 0.75; mats.append(m)              # sub-voxel shift
    m = ident.copy(); m[3], m[7], m[11] = 60, 0, 0; mats.append(m)                      # everything out of bounds
    m = ident.copy(); m[3], m[7], m[11] = -0.5, -0.5, -0.5; mats.append(m)              # on the window edge
    mats.append(ro.dof9tomatrix(np.array([0, 1.5, -2.0, 0.5, 3.0, -2.0, 4.0, 1.02, 0.97, 1.05], dtype=f32), 9))
    m = rng.normal(0, 0.02, 12).astype(f32) + ident; m[3], m[7], m[11] = 1.0, 2.0, -1.0; mats.append(m)
    # rim voxels: texture coordinates t = b + 0.5 in (0, 0.5] on the low faces / in [n - 0.5, n) on the high faces -- inside
    # the window, interpolating between an edge voxel and its clamped replica (include/cukernel.cuh:545, 515)
    m = ident.copy(); m[3], m[7], m[11] = -0.25, -0.3, -0.4; mats.append(m)             # low rim: t = 0.25, 0.2, 0.1
    m = ident.copy(); m[3], m[7], m[11] = 0.25, 0.3, 0.4; mats.append(m)                # high rim: t = n - 0.25 ...
    m = ident.copy(); m[0], m[5], m[10] = f32(47.5) / f32(47), f32(39.5) / f32(39), f32(23.5) / f32(23)
    m[3], m[7], m[11] = -0.25, -0.25, -0.25; mats.append(m)                             # both rims at once: b runs -0.25 .. n - 0.75
    mats = np.stack(mats)
    resampled = np.stack([ro.affine_transform(src, m, tgt.shape) for m in mats])
    sums = np.array([ro.corr_sums(tgt, src, m) for m in mats], dtype=np.float64)
    # different output box (imresize3d-style diagonal matrix)
    out_shape = (30, 50, 40)
    md = np.zeros(12, dtype=f32)
    md[0], md[5], md[10] = f32(48) / f32(40), f32(40) / f32(50), f32(24) / f32(30)
    save("affine.npz", src=src, tgt=tgt, mats=mats, resampled=resampled, corr_sums=sums,
         resize_matrix=md, resized=ro.affine_transform(src, md, out_shape))


def matrix_tables():
    rng = np.random.default_rng(3)
    p9 = rng.normal(0, 3, (8, 10)).astype(f32)
    p9[:, 7:] = 1 + rng.normal(0, 0.05, (8, 3)).astype(f32)
    dof = np.array([3, 6, 7, 9, 3, 6, 7, 9], dtype=np.int32)
    m9 = np.stack([ro.dof9tomatrix(p, int(d)) for p, d in zip(p9, dof)])
    a = rng.normal(0, 1, (5, 12)).astype(f32)
    b = rng.normal(0, 1, (5, 12)).astype(f32)
    prod = np.stack([ro.matrixmultiply(x, y) for x, y in zip(a, b)])
    rot = np.stack([ro.rot2matrix(f32(0.7), 50, 36, 20, ax) for ax in (1, 2, 3)])
    check_in = np.array([[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0],
                         [0.4, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0],
                         [1.3, 0, 0, 0, 0, 1.3, 0, 0, 0, 0, 1.39, 0],
                         [1.3, 0, 0, 0, 0, 1.3, 0, 0, 0, 0, 1.41, 0],
                         [1, 0, 0, 41, 0, 1, 0, 0, 0, 0, 1, 0],
                         [1, 0, 0, 39, 0, 1, 0, -28, 0, 0, 1, 15.9],
                         [0.6, 0, 0, 0, 0, 0.6, 0, 0, 0, 0, 0.6, 0]], dtype=f32)
    check_out = np.array([ro.checkmatrix(m, 50, 36, 20) for m in check_in])
    sizes = np.array([1, 15, 16, 17, 20, 36, 50, 64, 65, 100, 127, 128, 129, 130, 192, 200, 256, 257, 300, 320, 500,
                      512, 513, 1000, 1024], dtype=np.int32)
    snapped = np.array([do.snap_transform_size(int(s)) for s in sizes], dtype=np.int32)
    save("tables.npz", p9=p9, dof=dof, m9=m9, mm_a=a, mm_b=b, mm_out=prod, rot=rot, check_in=check_in,
         check_out=check_out, snap_in=sizes, snap_out=snapped)


def powell_ref():
    if not po.ref_available():
        raise SystemExit("oracle/_ref/libpowell_ref.so missing: run `make -C oracle ref` first")
    out = {}
    for kind, n in (("quad", 6), ("rosen", 6), ("rosen", 12), ("ridge", 9), ("zncc", 12)):
        func, counter, trace = make_cost(kind, n)
        p, xi = start_point(n)
        it, fret = po.ref_powell(p, xi, n, 1e-4, func, counter, 3000)
        key = f"{kind}{n}"
        out[key + "_p"] = p[1:].copy()
        out[key + "_xi"] = xi[1:, 1:].copy()
        out[key + "_fret"] = f32(fret)
        out[key + "_evals"] = np.int32(counter[0])
        out[key + "_trace"] = np.array(trace, dtype=f32)
    save("powell_ref.npz", **out)


def registration_case():
    """End-to-end reg3d_affine1 with the COMPILED REFERENCE Powell driving the oracle cost."""
    tgt = do.blob_volume((24, 32, 40), n_blobs=14, seed=9)
    known = ro.dof9tomatrix(np.array([0, 1.2, -0.8, 0.6, 2.0, -1.5, 1.0, 1.0, 1.0, 1.0], dtype=f32), 6)
    src = ro.affine_transform(tgt, known, tgt.shape)  # source = target warped by `known`
    powell = po.ref_powell if po.ref_available() else po.powell
    drv = ro.Reg3dAffine(powell)
    out = {}
    for method in (2, 5, 6, 7):
        reg, tmx, rec = drv.run(tgt, src, method, ftol=1e-4, it_limit=3000)
        out[f"m{method}_tmx"] = tmx
        out[f"m{method}_records"] = rec
        out[f"m{method}_reg"] = reg
    save("reg3d.npz", tgt=tgt, src=src, known=known, **out)


def registration_trajectories():
    """reg3d_traj.npz: EVERY matrix reg3d_affine1 (src/api_subfunc.cu:2889-2961) evaluates, with the cost it saw, for the cases of
    reg3d.npz -- affMethod 2, 5, 6, 7 from the identity, and the two input-matrix paths (affMethod 2: pre-transform, register
    from the identity, compose; affMethod 5: the matrix is the start point) -- driven by the COMPILED reference powell().  The
    -m gpu tests demand the same evaluations bit for bit (mil_reg3d_trace_*), not just an equally good optimum."""
    g = np.load(os.path.join(HERE, "reg3d.npz"))
    tgt, src = g["tgt"], g["src"]
    if not po.ref_available():
        raise SystemExit("oracle/_ref/libpowell_ref.so missing: run `make -C oracle ref` first")
    start = ro.dof9tomatrix(np.array([0, -1.0, 0.5, -0.5, 0, 0, 0, 1, 1, 1], dtype=f32), 3)
    real_corrfunc = ro.corrfunc
    out = {"start": start}
    cases = [(f"m{m}", m, False, None) for m in (2, 5, 6, 7)] + [("in2", 2, True, start), ("in5", 5, True, start)]
    for key, method, flag, tmx0 in cases:
        trace = []

        def tracing(target, sd_t, s, m):
            c = real_corrfunc(target, sd_t, s, m)
            trace.append(np.concatenate([np.asarray(m, dtype=f32), [f32(-c)]]).astype(f32))
            return c
        ro.corrfunc = tracing
        try:
            reg, tmx, rec = ro.Reg3dAffine(po.ref_powell).run(tgt, src, method, flag_tmx=flag, i_tmx=tmx0, ftol=1e-4, it_limit=3000)
        finally:
            ro.corrfunc = real_corrfunc
        if not flag:  # the same run as reg3d.npz holds
            assert np.array_equal(rec, g[f"m{method}_records"]) and np.array_equal(tmx, g[f"m{method}_tmx"])
        out[key + "_trace"], out[key + "_tmx"], out[key + "_records"], out[key + "_reg"] = np.stack(trace), tmx, rec, reg
        print(key, len(trace), "evaluations, ZNCC", rec[1], "->", rec[3])
    save("reg3d_traj.npz", **out)


if __name__ == "__main__":
    if "traj" in sys.argv[1:]:
        registration_trajectories()
        raise SystemExit(0)
    decon_cases()
    geometry_cases()
    affine_cases()
    matrix_tables()
    powell_ref()
    registration_case()
    registration_trajectories()
