"""The CuPy-array surface of the drop-in classes (the reference returns ``cupy.ndarray``,
/root/reference/tomobar/methodsIR_CuPy.py:484, regularisersCuPy.py:64-65,198-199) driven with a stand-in ``cupy`` module
(tests/_cupy_standin.py -- CuPy-on-ROCm is not installed here; the product only asks for the array's module name and speaks
DLPack, so the stand-in exercises exactly the branches a real ``cupy.ndarray`` takes): a caller that hands CuPy arrays in
gets CuPy arrays back, nothing is copied on the way in, the caller's array is never written, and the values are the torch
path's bit for bit."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _cupy_standin  # noqa: E402


def test_array_library_plumbing_on_host_tensors(monkeypatch):
    """ops.is_cupy / base_ptr / like need no GPU (DLPack works on host tensors too): the return-type rule itself."""
    cupy = _cupy_standin.install(monkeypatch)
    from tomobar_amd import ops
    t = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    c = cupy.ndarray(t)
    assert ops.is_cupy(c) and not ops.is_cupy(t) and not ops.is_cupy(np.zeros(3)) and not ops.is_cupy(None)
    assert ops.base_ptr(c) == t.data_ptr() == ops.base_ptr(t) and ops.base_ptr(np.zeros(3)) is None
    res = torch.full((2, 2), 7.0)
    back = ops.like(res, c)
    assert type(back) is cupy.ndarray and type(back).__module__ == "cupy"
    assert back.data.ptr == res.data_ptr(), "cupy.from_dlpack must wrap the result, not copy it"
    assert ops.like(res, t) is res and ops.like(res, np.zeros(3)) is res and ops.like(res, None) is res
    assert ops.like("not a tensor", c) == "not a tensor"


@pytest.mark.gpu
def test_cupy_in_cupy_out_zero_copy(monkeypatch, oracle):
    cupy = _cupy_standin.install(monkeypatch)
    from tomobar_amd import ops
    from tomobar_amd.methodsDIR_CuPy import RecToolsDIRCuPy
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    from tomobar_amd.regularisersCuPy import PD_TV_cupy, ROF_TV_cupy
    rng = np.random.default_rng(5)
    nz, n, na = 6, 48, 40
    angles = np.linspace(0, np.pi, na, endpoint=False)
    sino_t = torch.from_numpy(rng.random((nz, na, n)).astype(np.float32)).cuda()
    keep = sino_t.clone()
    sino_c = cupy.ndarray(sino_t)

    # ---- ops.to_device: a CuPy array is taken over through DLPack without a copy
    got = ops.to_device(sino_c, 0)
    assert isinstance(got, torch.Tensor) and got.data_ptr() == sino_t.data_ptr() == ops.base_ptr(sino_c)

    # ---- RecToolsIRCuPy.FISTA / ADMM / powermethod: CuPy in -> CuPy out, same values as the torch path
    def data(x):
        return {"projection_data": x, "data_axes_labels_order": ["detY", "angles", "detX"]}

    rt = RecToolsIRCuPy(n, 0, nz, 0.0, angles, n, 0, 4)
    algo = {"iterations": 3, "lipschitz_const": 3000.0, "nonnegativity": True}
    reg = {"method": "PD_TV", "regul_param": 1e-3, "iterations": 7}
    for call, a, r in ((rt.FISTA, algo, reg), (rt.FISTA, algo, None), (rt.ADMM, dict(algo), {"method": "ROF_TV", "regul_param": 1e-3, "iterations": 5})):
        want = call(data(sino_t), dict(a), None if r is None else dict(r))
        res = call(data(sino_c), dict(a), None if r is None else dict(r))
        assert isinstance(want, torch.Tensor)
        assert type(res) is cupy.ndarray and res.shape == (nz, n, n) and res.dtype == np.float32
        assert np.array_equal(res.get(), want.cpu().numpy())
        assert torch.equal(sino_t, keep), "the caller's projection data must not be written"

    # ---- the regularisers: result type follows the input, the output never aliases the input
    vol_t = torch.from_numpy(rng.random((5, 20, 33)).astype(np.float32)).cuda()
    vol_c = cupy.ndarray(vol_t)
    for fn, args in ((PD_TV_cupy, (0.02, 9, 0, 1, 12.0, 0, False)), (ROF_TV_cupy, (0.02, 6, 0.002, 0, False))):
        want = fn(vol_t, *args)
        res = fn(vol_c, *args)
        assert type(res) is cupy.ndarray and res.data.ptr != vol_t.data_ptr()
        assert np.array_equal(res.get(), want.cpu().numpy())
    img_c = cupy.ndarray(vol_t[2].contiguous())          # 2D input: same rule
    assert type(PD_TV_cupy(img_c, 0.02, 4)) is cupy.ndarray
    assert isinstance(PD_TV_cupy(vol_t.cpu().numpy(), 0.02, 4), torch.Tensor)   # numpy in -> device tensor out

    # ---- RecToolsDIRCuPy: FBP works in place on its own copy (the caller's CuPy array is left alone), FORWPROJ / BACKPROJ
    dr = RecToolsDIRCuPy(n, 0, nz, 0.0, angles, n, device_projector=0)
    proj_c = cupy.ndarray(sino_t.permute(1, 0, 2).contiguous())   # ["angles", "detY", "detX"], FBP's own order: no swap copy
    before = proj_c.get().copy()
    rec_c = dr.FBP(proj_c)
    rec_t = dr.FBP(torch.from_numpy(before).cuda())
    assert type(rec_c) is cupy.ndarray and isinstance(rec_t, torch.Tensor)
    assert np.array_equal(rec_c.get(), rec_t.cpu().numpy())
    assert np.array_equal(proj_c.get(), before), "FBP filtered the caller's array in place"
    fwd = dr.FORWPROJ(cupy.ndarray(rec_t))
    assert type(fwd) is cupy.ndarray and fwd.shape == (nz, na, n)
    bck = dr.BACKPROJ(fwd)
    assert type(bck) is cupy.ndarray and bck.shape == (nz, n, n)
    assert np.array_equal(bck.get(), dr.BACKPROJ(torch.from_dlpack(fwd)).cpu().numpy())


@pytest.mark.gpu
def test_cupy_caller_on_its_own_stream(monkeypatch):
    """INTEGRATION.md, stream rule: the classes order their work on torch's CURRENT stream.  A CuPy caller that computes on
    a non-default stream hands that stream to torch (``torch.cuda.ExternalStream(cupy_stream.ptr)``) -- then the library's
    launches are ordered behind the caller's own kernels without any synchronisation."""
    cupy = _cupy_standin.install(monkeypatch)
    from tomobar_amd.regularisersCuPy import PD_TV_cupy
    side = torch.cuda.Stream()                                  # stands for cupy.cuda.Stream(non_blocking=True)
    ext = torch.cuda.ExternalStream(side.cuda_stream)           # what the caller builds from cupy_stream.ptr
    vol = torch.rand((12, 64, 64), device="cuda")
    want = PD_TV_cupy(vol * 2.0, 0.05, 6)
    torch.cuda.synchronize()
    with torch.cuda.stream(ext):
        x = vol * 2.0                                           # the caller's own kernel, queued on its stream ...
        got = PD_TV_cupy(cupy.ndarray(x), 0.05, 6)              # ... and the proximal step right behind it, no sync between
    ext.synchronize()
    assert type(got) is cupy.ndarray and np.array_equal(got.get(), want.cpu().numpy())
