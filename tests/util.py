"""shared helpers for the parity tests"""
import numpy as np

from er3t_amd.scene import Scene, TARGET_FLUX, TARGET_RADIANCE


def slab_scene(tau=1.0, omega=1.0, apf=0.85, albedo=0.0, sza=30.0, nz=4, ztop=4000.0, nx=1, ny=1, nz3=0,
               target=TARGET_FLUX | TARGET_RADIANCE, vza=(0.0,), vaa=(0.0,), qmax=0.0, abs_tau=0.0, ang=None, pha=None,
               solver=0, dx=200.0, dy=200.0, wmin=0.2, wfac=1.0, le_cmin=0.0):
    """plane-parallel slab of total optical thickness `tau` spread over nz equal layers; with nz3 > 0 the lowest
    nz3 layers are carried by an (nx, ny, nz3) 3-D grid holding the same homogeneous medium instead of the 1-D profile.
    (le_cmin = 0: the known answers built on these scenes include identities that hold estimate by estimate -- a Lambertian
    surface under a vacuum -- which a roulette on the estimates' weights keeps in the mean only)"""
    zgrd = np.linspace(0.0, ztop, nz+1)
    dz = ztop/nz
    ext = np.full((1, nz), tau/ztop)
    omg = np.full((1, nz), omega)
    ap = np.full((1, nz), apf)
    kw = dict(zgrd=zgrd, omg1d=omg, apf1d=ap, abs1d=np.full(nz, abs_tau/ztop), nx=nx, ny=ny, dx=dx, dy=dy,
              sfc_mtype=1, sfc_param=[albedo, 0, 0, 0, 0], src_the=180.0-sza, src_phi=270.0, src_qmax=qmax,
              target=target, solver=solver, wmin=wmin, wfac=wfac, ang=ang, pha=pha, le_cmin=le_cmin)
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


def block_scene(kind, sza=45.0, nx=8, ny=6, d=100.0):
    """vacuum atmosphere of four 250 m layers with ONE voxel filled, column (2, 3) of the third layer (z = 500..750 m):
    kind 'absorber': opaque and black (extinction 100/m, omega 0) under a sun from -x at `sza` -> a sharp shadow on the ground;
    kind 'scatterer': thin, conservative, isotropic under an overhead sun -> a small cloud whose image shifts with the view"""
    nz, dz = 4, 250.0
    zgrd = np.arange(nz+1)*dz
    ext = np.zeros((1, 1, ny, nx), dtype=np.float32)
    ext[0, 0, 3, 2] = 100.0 if kind == 'absorber' else 4.0e-5
    kw = dict(zgrd=zgrd, ext1d=np.zeros((1, nz)), omg1d=np.ones((1, nz)), apf1d=-2.0*np.ones((1, nz)), abs1d=np.zeros(nz),
              nx=nx, ny=ny, dx=d, dy=d, nz3=1, iz3l=3, extp=ext, omgp=np.full_like(ext, 0.0 if kind == 'absorber' else 1.0),
              apfp=np.full_like(ext, -2.0), sfc_mtype=1, sfc_param=[0.0, 0, 0, 0, 0], src_qmax=0.0, src_phi=0.0, le_cmin=0.0)
    if kind == 'absorber':
        return Scene(src_the=180.0-sza, target=TARGET_FLUX, **kw)
    # views: nadir; 45 degrees with the light travelling towards +x (sensor on the +x side); the same towards -x
    return Scene(src_the=180.0, target=TARGET_RADIANCE, view_the=[180.0, 135.0, 135.0], view_phi=[0.0, 180.0, 0.0],
                 view_zloc=[705000.0]*3, nxr=nx, nyr=ny, **kw)


def block_expectations(nx=8, d=100.0):
    """what block_scene must give.  Shadow (sun at 45 deg from -x): a ray reaching the ground at x_s has passed through the
    voxel x in [2d, 3d], z in [500, 750] iff x_s in [2d+500, 3d+750] (mod nx d) -> fraction of the direct beam left per ground
    column of row 3.  Parallax (view at 45 deg, pixels registered at z = 0): an event at (x, z) is seen at x -/+ z; for a thin
    voxel lit from above events are uniform in it, so the image is the trapezoid U(2d,3d) -/+ U(500,750), folded into the domain"""
    L = nx*d
    xs = (np.arange(4000)+0.5)*L/4000.0
    blocked = ((xs-(2*d+500.0)) % L) <= (d+250.0)
    shadow = 1.0 - blocked.reshape(nx, -1).mean(axis=1)
    xx, zz = np.meshgrid(2*d+(np.arange(400)+0.5)*d/400.0, 500.0+(np.arange(400)+0.5)*250.0/400.0)
    image = {}
    for sign in (-1.0, 1.0):
        xr = (xx + sign*zz) % L
        image[sign] = np.histogram(xr.ravel(), bins=nx, range=(0.0, L))[0]/xx.size
    return shadow, image
