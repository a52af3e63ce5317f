"""shared helpers for the parity tests"""
import numpy as np

from er3t_amd.scene import Scene, TARGET_FLUX, TARGET_RADIANCE


def slab_scene(tau=1.0, omega=1.0, apf=0.85, albedo=0.0, sza=30.0, nz=4, ztop=4000.0, nx=1, ny=1, nz3=0,
               target=TARGET_FLUX | TARGET_RADIANCE, vza=(0.0,), vaa=(0.0,), qmax=0.0, abs_tau=0.0, ang=None, pha=None,
               solver=0, dx=200.0, dy=200.0, wmin=0.2, wfac=1.0):
    """plane-parallel slab of total optical thickness `tau` spread over nz equal layers; with nz3 > 0 the lowest
    nz3 layers are carried by an (nx, ny, nz3) 3-D grid holding the same homogeneous medium instead of the 1-D profile"""
    zgrd = np.linspace(0.0, ztop, nz+1)
    dz = ztop/nz
    ext = np.full((1, nz), tau/ztop)
    omg = np.full((1, nz), omega)
    ap = np.full((1, nz), apf)
    kw = dict(zgrd=zgrd, omg1d=omg, apf1d=ap, abs1d=np.full(nz, abs_tau/ztop), nx=nx, ny=ny, dx=dx, dy=dy,
              sfc_mtype=1, sfc_param=[albedo, 0, 0, 0, 0], src_the=180.0-sza, src_phi=270.0, src_qmax=qmax,
              target=target, solver=solver, wmin=wmin, wfac=wfac, ang=ang, pha=pha)
    if nz3 > 0:
        ext1 = ext.copy(); ext1[0, :nz3] = 0.0
        kw.update(ext1d=ext1, nz3=nz3, iz3l=1,
                  extp=np.full((1, nz3, ny, nx), tau/ztop, dtype=np.float32),
                  omgp=np.full((1, nz3, ny, nx), omega, dtype=np.float32),
                  apfp=np.full((1, nz3, ny, nx), apf, dtype=np.float32))
    else:
        kw.update(ext1d=ext)
    if target & TARGET_RADIANCE:
        vza = np.atleast_1d(vza).astype(float); vaa = np.resize(np.atleast_1d(vaa).astype(float), vza.size)
        kw.update(view_the=list(180.0-vza), view_phi=list((270.0-vaa) % 360.0), view_zloc=[705000.0]*vza.size, nxr=nx, nyr=ny)
    return Scene(**kw)


def batch_stats(run_fn, nbatch, nper, seed):
    """run `nbatch` independent batches (disjoint photon ids) -> mean over batches and standard error of that mean"""
    out = [run_fn(nper, seed, b*nper) for b in range(nbatch)]
    a = np.stack(out)
    return a.mean(axis=0), a.std(axis=0, ddof=1)/np.sqrt(nbatch)
