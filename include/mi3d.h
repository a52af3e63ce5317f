/*
 * mi3d.h — C-ABI of libmi3drt.so, the MI355X-native 3D Monte-Carlo radiative-transfer solver.
 *
 * This library replaces the photon-transport program that the reference toolbox (hong-chen/er3t)
 * launches as a subprocess.  The reference has no FFI for this path; its interface is
 *
 *     os.system("<MCARATS_V010_EXE> <Nphoton> <solver 0|1|2> <inp.txt> <out.bin>")
 *                                         (er3t/rtm/mca/mca_run.py:101-115,179-181)
 *
 * where <inp.txt> is a Fortran namelist (er3t/rtm/mca/mca_inp.py:15-384,636-697) naming three
 * little-endian float32 side files (er3t/rtm/mca/mca_atm.py:373-389, mca_sca.py:82-92,
 * mca_sfc.py:136-146) and <out.bin> is a float32 Fortran-order dump described by a GrADS .ctl
 * (er3t/rtm/mca/mca_out.py:48-103).  Every entry point below states which piece of that contract
 * it replaces.  Host arrays are given in exactly the layout of the reference's side files, so a
 * caller can hand over `np.fromfile(side_file, '<f4')` unchanged.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MI3D_E* code; the message for the calling
 *     thread's last failure is available from mi3d_last_error().
 *   - host input buffers are owned by the caller and only read during the call (the library
 *     copies them to the device).  Output buffers are caller-allocated.
 *   - a handle is not re-entrant; different handles may be used from different threads.
 *   - angles are degrees, lengths metres, in the solver's own conventions (Src_the = 180 - SZA,
 *     azimuth = direction of photon travel / camera pointing, counter-clockwise from +x = east;
 *     er3t/rtm/mca/mcarats.py:374-383,527-549).
 */
#ifndef MI3D_H
#define MI3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI3D_VERSION 100 /* 0.1.0 */

/* error codes */
#define MI3D_OK 0
#define MI3D_EINVAL (-1)   /* bad argument / inconsistent shapes            */
#define MI3D_ESTATE (-2)   /* call order (e.g. run before set_atm1d)        */
#define MI3D_EDEVICE (-3)  /* HIP runtime failure (no GPU, OOM, launch)     */
#define MI3D_EUNSUP (-4)   /* valid MCARaTS option this solver does not do  */

/* limits (compile-time table sizes in the kernels) */
#define MI3D_MAX_NP1D 4    /* scattering components of the 1-D background   */
#define MI3D_MAX_NP3D 4    /* scattering components of the 3-D region       */
#define MI3D_MAX_VIEW 16   /* radiance view directions per launch           */
#define MI3D_NCOUNTER 24   /* length of the counter vector, see below       */

/* target (Wld_mtarget, er3t/rtm/mca/mcarats.py:267-287) */
#define MI3D_TARGET_FLUX 1
#define MI3D_TARGET_RADIANCE 2
#define MI3D_TARGET_HEAT 4 /* heating rates beside the fluxes: Flx_mhrt = 1 (mcarats.py:279-283; mca_inp.py:124); with MI3D_TARGET_FLUX */

/* solver (2nd CLI argument of the reference's command line, mcarats.py:450-454) */
#define MI3D_SOLVER_3D 0
#define MI3D_SOLVER_P3D 1 /* partial 3-D: 3-D direct beam, independent columns for all scattered light */
#define MI3D_SOLVER_IPA 2

/* surface model ids (Sfc_mtype / jsfc2d, er3t/rtm/mca/mca_sfc.py:94-128) */
#define MI3D_SFC_LAMBERT 1
#define MI3D_SFC_DSM 2 /* diffuse-specular mixture (whitecaps + Cox-Munk facets): five parameters */
#define MI3D_SFC_LSRT 4

/* indices into the counter vector returned by mi3d_get_counters (all uint64, summed over
 * every run since the last mi3d_reset).  These are the measured quantities SURVEY.md §8(d)
 * builds the algorithmic-bytes figure from. */
#define MI3D_CNT_PHOTONS 0      /* photon histories completed                              */
#define MI3D_CNT_STEPS 1        /* cell-boundary steps + collision stops during transport  */
#define MI3D_CNT_STEPS3D 2      /* ... of which inside the 3-D region (one ext read each)  */
#define MI3D_CNT_SCATTER 3      /* scattering collisions                                   */
#define MI3D_CNT_SURFACE 4      /* surface reflections                                     */
#define MI3D_CNT_LE_RAYS 5      /* local-estimate rays cast (events x views)               */
#define MI3D_CNT_LE_STEPS 6     /* cell steps walked by local-estimate rays                */
#define MI3D_CNT_LE_STEPS3D 7   /* ... of which inside the 3-D region                      */
#define MI3D_CNT_LE_COLUMN 8    /* LE rays answered from the vertical optical-depth table  */
#define MI3D_CNT_FLUX_TALLY 9   /* flux plane crossings tallied                            */
#define MI3D_CNT_ROULETTE 10    /* Russian-roulette games played                           */
#define MI3D_CNT_KILLED 11      /* histories ended by roulette                             */
#define MI3D_CNT_ESCAPED 12     /* histories that left through the top                     */
#define MI3D_CNT_ABSORBED 13    /* histories ended with zero weight (black surface/voxel)  */
#define MI3D_CNT_SCHED_A_LANES 14 /* scheduler diagnostics: lanes that stepped / lane slots offered */
#define MI3D_CNT_SCHED_A_SLOTS 15
#define MI3D_CNT_SCHED_B_LANES 16 /* lanes with an event pending / lane slots of event passes       */
#define MI3D_CNT_SCHED_B_SLOTS 17
#define MI3D_CNT_TICKS_A 18     /* wave clock ticks / 64, summed over lanes, spent in: voxel steps,     */
#define MI3D_CNT_TICKS_B0 19    /* uniform-layer runs,                                                  */
#define MI3D_CNT_TICKS_B12 20   /* events and their tallies,                                            */
#define MI3D_CNT_TICKS_B34 21   /* starting marched views and handing out photon ids,                   */
#define MI3D_CNT_TICKS_B5 22    /* the finish block (new direction),                                    */
#define MI3D_CNT_TICKS_B6 23    /* the Philox block                                                     */

typedef struct mi3d_solver mi3d_solver;

int mi3d_version(void);
const char *mi3d_last_error(void);

/* Number of visible HIP devices (0 if none / runtime unusable). Does not create a context. */
int mi3d_device_count(void);

/* Create a solver bound to HIP device `device`.  Replaces process start-up of the reference's
 * solver executable (mca_run.py:179-181).  Fails with MI3D_EDEVICE when no GPU is usable: there
 * is no CPU fallback. */
int mi3d_create(int device, mi3d_solver **out);
int mi3d_destroy(mi3d_solver *h);

/* 1-D background atmosphere = namelist keys Atm_nz, Atm_zgrd0, Atm_np1d, Atm_ext1d(1:,ip),
 * Atm_omg1d(1:,ip), Atm_apf1d(1:,ip), Atm_abs1d(1:,1)   (er3t/rtm/mca/mca_atm.py:68-139).
 *   zgrd[nz+1]  level heights, ascending, zgrd[0] is the surface
 *   ext/omg/apf [np1d][nz] (component-major, layer index fastest), abs[nz]
 * apf selects the phase function of a component (mca_atm.py:101,262,276; rtm/mca/util.py:153):
 *   apf <= -1.5 isotropic; -1.5 < apf <= -1 Rayleigh; -1 < apf < 1 Henyey-Greenstein with g=apf;
 *   apf >= 1 tabulated, 1-based real table index (fraction = mix of the two neighbours). */
int mi3d_set_atm1d(mi3d_solver *h, int nz, const double *zgrd, int np1d, const float *ext,
                   const float *omg, const float *apf, const float *abs);

/* 3-D region = keys Atm_nx, Atm_ny, Atm_dx, Atm_dy, Atm_nz3, Atm_iz3l, Atm_np3d and the side file
 * Atm_inpfile (er3t/rtm/mca/mca_atm.py:231-337,373-389).  Arrays are in the FILE layout: x
 * fastest, then y, then z; component blocks one after another:
 *   abst[nz3][ny][nx]  gas-absorption perturbation added to abs1d of the layer (may be NULL = 0)
 *   extp/omgp/apfp [np3d][nz3][ny][nx]
 * iz3l is the 1-based index of the lowest 3-D layer exactly as the namelist carries it.
 * Atm_tmpa3d (temperature perturbation) is not needed for solar transport and is not passed.
 * nz3 == 0 removes the 3-D region (then nx, ny give the horizontal tally grid only). */
int mi3d_set_atm3d(mi3d_solver *h, int nx, int ny, int nz3, int iz3l, int np3d, double dx,
                   double dy, const float *abst, const float *extp, const float *omgp,
                   const float *apfp);

/* Tabulated phase functions = keys Sca_npf, Sca_nangi and the side file Sca_inpfile
 * (er3t/rtm/mca/mca_sca.py:72-92): ang[nang] degrees ascending from 0 to 180, pha[npf][nang].
 * Tables are renormalised to (1/2)∫P dμ = 1 and treated as piecewise linear in μ = cos(angle).
 * npf == 0 clears the tables. */
int mi3d_set_phase(mi3d_solver *h, int nang, int npf, const float *ang, const float *pha);

/* Uniform surface = keys Sfc_mtype, Sfc_param(1:5) (er3t/rtm/mca/mcarats.py:393-399). */
int mi3d_set_surface(mi3d_solver *h, int mtype, const float param[5]);

/* 2-D surface = keys Sfc_nxb, Sfc_nyb and the side file Sfc_inpfile
 * (er3t/rtm/mca/mca_sfc.py:81-146), file layout: tmps[nyb][nxb] (ignored, may be NULL),
 * jsfc[nyb][nxb] (model id stored as float), psfc[5][nyb][nxb]. */
int mi3d_set_surface2d(mi3d_solver *h, int nxb, int nyb, const float *tmps, const float *jsfc,
                       const float *psfc);

/* Solar source = keys Src_flx, Src_qmax, Src_the, Src_phi (er3t/rtm/mca/mcarats.py:374-383). */
int mi3d_set_source(mi3d_solver *h, double flx, double qmax_deg, double the_deg, double phi_deg);

/* Radiance views = keys Rad_nrad, Rad_the, Rad_phi, Rad_zloc, Rad_zref, Rad_nxr, Rad_nyr for
 * Rad_mrkind = 2 (pixel-averaged radiance; er3t/rtm/mca/mcarats.py:285-307,360-367).  The
 * reference passes one view per solver process; this library takes up to MI3D_MAX_VIEW per
 * launch.  the_deg > 90: down-looking sensor (180 = nadir; Rad_the = 180 - sensor_zenith_angle,
 * mcarats.py:305): zloc[] is the sensor height, radiance is collected where the line of sight
 * crosses min(zloc, top of atmosphere) and registered to the pixel where it meets z = zref.
 * the_deg < 90: up-looking sensor (0 = zenith; "looking up, 180 deg straight up",
 * mcarats.py:499-502): light travelling down to a plane of sensors at max(zloc, surface), from
 * events above it only, registered where the line of sight meets that plane.  the_deg = 90
 * (horizontal) is rejected. */
int mi3d_set_views(mi3d_solver *h, int nview, const double *the_deg, const double *phi_deg,
                   const double *zloc, double zref, int nxr, int nyr);

/* All-sky cameras: Rad_mrkind = 1, "local radiance averaged over solid angle" (er3t/rtm/mca/mca_inp.py:141-144; set by
 * er3t/rtm/mca/mcarats.py:291-296, 369-372 for sensor_type = 'all-sky': Rad_qmax = 178, Rad_apsize = 0.05, Rad_xpos/ypos,
 * a 500 x 500 image).  Replaces mi3d_set_views for the job (a job has one kind of radiance).  Camera i stands at
 * (xpos[i] Lx, ypos[i] Ly, zloc[i]); its axes are the world axes turned by the Z-Y-Z rotations phi, the, psi (Rad_phi,
 * Rad_the, Rad_psi, mca_inp.py:324-330) and it looks along its z axis (the = 180: straight down, the = 0: straight up, as
 * for the satellite views); it sees directions within qmax/2 degrees of that axis (Rad_qmax: full angle of the cone).  Pixel
 * map: polar, Rad_mpmap = 1: a direction at angle theta from the axis and azimuth phi about it falls at U = theta cos(phi),
 * V = theta sin(phi), the image spanning umax x vmax degrees (Rad_umax, Rad_vmax) in nxr x nyr pixels.  Estimator: every
 * collision and reflection sends w P / (4 pi) exp(-tau) / r^2 (surface: w R cos / pi ...) to the nearest periodic image of
 * the camera, r not counted below apsize metres (Rad_apsize); 3-D solver only.  MCARaTS' own regularisation of the 1/r^2
 * estimator (Rad_difr*, Rad_rmin0 ...) is not in the reference tree and not applied: see DESIGN.md. */
int mi3d_set_cameras(mi3d_solver *h, int ncam, const double *the_deg, const double *phi_deg, const double *psi_deg,
                     const double *xpos, const double *ypos, const double *zloc, const double *qmax_deg,
                     const double *umax_deg, const double *vmax_deg, const double *apsize, int nxr, int nyr);

/* Job options = 1st/2nd CLI arguments and keys Wld_mtarget, Flx_mflx, Pho_wmin
 * (er3t/rtm/mca/mcarats.py:267-287,450-454; er3t/rtm/mca/mca_inp.py:196-198).
 *   target   MI3D_TARGET_FLUX | MI3D_TARGET_RADIANCE (bit-or of both is allowed); MI3D_TARGET_FLUX | MI3D_TARGET_HEAT is the
 *            reference's target='heating rate' (Flx_mflx = 3, Flx_mhrt = 1): the weight every collision takes from a photon --
 *            gas absorption and the absorbing part of every constituent -- is tallied in the cell of the collision
 *   solver   MI3D_SOLVER_3D | MI3D_SOLVER_P3D | MI3D_SOLVER_IPA
 *   wmin     Russian-roulette weight threshold (Pho_wmin, default 0.2)
 *   wfac     weight survivors of the roulette continue with (Pho_wfac, default 1); survival probability w/wfac
 *   column_le  1: answer exactly vertical views from a per-column optical-depth table (exact,
 *              one read per event); 0: always march the local-estimate ray cell by cell. */
int mi3d_set_options(mi3d_solver *h, int target, int solver, double wmin, double wfac, int column_le);

/* Russian roulette on marched local-estimate rays (a variance-reduction option of this solver, unbiased; no namelist key:
 * MCARaTS' own `Rad_difr*` / truncation options are different devices).  tau1 > 0: a ray survives to optical depth
 * tau > tau1 with probability exp(-(tau - tau1)) and then contributes exp(-tau1) instead of exp(-tau), so that rays from
 * deep inside a cloud stop after about tau1 + 1 optical depths instead of being marched to the cut-off at 16.  Exactly
 * vertical views of a sensor above the atmosphere (answered from the column table) are never affected.  tau1 = 0
 * (default): off. */
int mi3d_set_le_roulette(mi3d_solver *h, double tau1);
/* Russian roulette on the WEIGHT of marched local-estimate rays of satellite views (Rad_mrkind = 2; unbiased; no namelist key).  A
 * local estimate carries c = w P(angle towards the sensor) / 4 pi (a reflection: w R cos / pi); with a forward-peaked phase function
 * (Henyey-Greenstein g = 0.85: P between 0.05 and 80) most of them carry a few per cent of what the few near the peak carry, and
 * the noise of a pixel is made by the latter.  cmin > 0: a ray with c < cmin is marched with probability c / cmin and then carries
 * cmin.  The mean is untouched; fewer than half the rays are marched.  Views answered from the column table (mi3d_set_options
 * column_le) are never affected: their estimates cost nothing.  cmin = 0 (default): off. */
int mi3d_set_le_weight_roulette(mi3d_solver *h, double cmin);

/* Select the instrumented build of the transport kernel, which fills every MI3D_CNT_* counter
 * (the default build only counts MI3D_CNT_PHOTONS and is the one to time).  The counters are a
 * deterministic function of (scene, seed, photon ids), so measuring them on a sub-sample of the
 * job's photon ids gives the job's per-photon averages. */
int mi3d_set_counting(mi3d_solver *h, int on);

/* Bind caller-owned DEVICE buffers for the raw tallies (so that a host framework can all-reduce
 * them in place with RCCL) and the HIP stream to launch on.  Any pointer may be NULL: the library
 * then keeps its own buffer / uses the null stream.  Sizes (FLOAT64 elements: a float32 accumulator
 * stops growing once it exceeds 2^24 contributions' worth, which one pixel of a small grid reaches within
 * a few million photons):
 *   rad_sum  [nview][nyr][nxr]
 *   flux_sum [3][nz+1][ny][nx]      RAW planes: direct-down, DIFFUSE-down, up (one atomic per level
 *                                    crossing); mi3d_get_flux / mi3d_stats_add form total-down = direct +
 *                                    diffuse, so only sums of raw buffers (an all-reduce) are meaningful */
int mi3d_bind_device_buffers(mi3d_solver *h, void *rad_sum, void *flux_sum, void *stream);
/* ... and for the heating-rate tally (MI3D_TARGET_HEAT): heat_sum [nz][ny][nx] float64, weight absorbed per cell; NULL: the
 * library's own buffer. */
int mi3d_bind_heating_buffer(mi3d_solver *h, void *heat_sum);

/* Build the device-side scene (layout transform, total extinction, column optical depth,
 * phase-function CDFs).  Called implicitly by mi3d_run when inputs changed; exposed so that
 * set-up can be excluded from the timed region. */
int mi3d_prepare(mi3d_solver *h);

/* Zero tallies and counters. */
int mi3d_reset(mi3d_solver *h);

/* Transport `nphoton` photon histories with global ids [photon_offset, photon_offset+nphoton)
 * of the random stream keyed by `seed` (Wld_jseed), accumulating into the tallies.  Asynchronous.
 * COMPLETION RULE (one, since round 6): when mi3d_run returns, everything it has started is queued on the handle's main stream -- the stream
 * bound with mi3d_bind_device_buffers, else the handle's own -- or joined to it.  Work the caller queues on that stream afterwards (a copy out
 * of a bound buffer, an all-reduce) and every mi3d call finds the tallies complete in stream order; hipStreamSynchronize of that stream, or
 * mi3d_sync, makes the host wait.  (tests/test_lib_abi.py holds it with a raw copy out of a bound buffer.)  The only exception is asked for
 * by name: mi3d_set_tuning "overlap_sort" 2, flux jobs on the handle's own buffers and stream -- the last record sort of a run may then still
 * be on its way on a stream of the handle's own; every mi3d call that reads, clears or re-homes tallies joins it first.
 * Replaces the reference solver's main loop ("<exe> <Nphoton> <solver> <inp> <out>", mca_run.py:113). */
int mi3d_run(mi3d_solver *h, uint64_t nphoton, uint64_t seed, uint64_t photon_offset);

/* Wait for outstanding launches.  A job whose marched views go through event lists (mi3d_last_kernel: "... + k_rays") does not
 * make mi3d_run wait for its last launches: whether one of their lists ran full -- the run's tallies are then incomplete,
 * MI3D_ESTATE -- is reported by the call that looks at the tallies next: mi3d_sync, mi3d_get_radiance, mi3d_get_counters,
 * mi3d_stats_add (it reads the job's tallies into the run field), mi3d_stats_end_run, mi3d_stats_get, or the next mi3d_run.  A caller that reads bound device buffers
 * itself calls mi3d_sync first and checks its return value.  mi3d_reset forgets the runs before it. */
int mi3d_sync(mi3d_solver *h);
/* Name of the transport kernel build that served the last mi3d_run of this handle ("k_transport_lean<COUNT,P3D,0,MIX>": the lean
 * build for radiance answered from the column table, "k_transport_lean<COUNT,P3D,2,MIX> + k_rays": marched views through event
 * records, "k_transport_flux<COUNT,P3D,MIX> + k_tl_scatter + k_tl_sum": flux jobs -- MIX 0: one 1-D and one 3-D constituent with
 * analytic phase functions (er3t's default scene), 1: a second 3-D constituent, 2: the general mixture (several 1-D constituents,
 * tabulated phase functions staged in LDS) --, "k_transport<COUNT,MARCH,FLUX,P3D>": the general kernel (flux together with radiance,
 * more than two 3-D constituents, tables too large for the LDS, kernel choice 1); "" before the first launch).  For logs and
 * measurements (bench.py, profiles/): results do not depend on it. */
const char *mi3d_last_kernel(mi3d_solver *h);
/* Which build of the transport kernel may serve a launch: 0 (default) the lean ones wherever they apply -- marched satellite
 * views through the ray kernel (k_transport_lean<.,.,2> + k_rays) --, 1 always the general one (k_transport).  Both implement the
 * same function photon id -> history and the same estimator; the choice is for A/B measurements and for the parity tests, which
 * hold every build against the oracle on the same scene.  The environment variable MI3D_KERNEL=generic sets the
 * default of new handles.  (The ray kernel keeps one event list per XCD in device memory, sized from a pilot launch and never
 * beyond a quarter of the memory free at the time -- launches are sized to the lists, so small lists cost launches, not results; when
 * not even 65 536 records per list are to be had the library warns and the general kernel serves the job.  Choice 2 of rounds 2-4, the
 * lean loop with the rays of marched views walked inside it, was retired in round 5: MI3D_EINVAL.) */
int mi3d_set_kernel(mi3d_solver *h, int choice);
/* Tuning knobs of the launch machinery, for measurements and for tests that must reach its corners at small sizes.  None changes a
 * result beyond the order of float64 sums; a caller of the drop-in path needs none of them.  Unknown key or value out of range: MI3D_EINVAL.
 * An environment variable of the same name in capitals with the prefix MI3D_ (MI3D_TILE_COLS ...) sets the default of new handles where
 * the last column says so.
 *
 *   key            default  range      what                                                                          set by (log)                        env
 *   -------------  -------  ---------  ----------------------------------------------------------------------------  ----------------------------------  ---
 *   tile_cols      -1       -1..4096   tile edge of the photon order in columns; 0: id order, -1: chosen from scene   profiles/r02/tile_sweep_les480.log  yes
 *   batch_log2     30       8..30      most photons per launch                                                       profiles/r05/ab_batch_2p30.log      yes
 *   evcap_log2     28       10..28     records per event list (marched views); never > 1/4 of the free memory         profiles/r05/ab_evcap.log           yes
 *   tlcap_log2     31       16..31     records per tally-record list (flux jobs); small values: tests of full lists   tests/test_gpu_parity.py            no
 *   rad_spread     -1       -1, 0, 1   accumulation image with one pixel per 128-byte line; -1: where the photon     profiles/r04/ab_rad_line_density    yes
 *                                      loop itself tallies by atomics
 *   rad_row_pad    -1       -1..4096   padding of that image's rows in pixels; -1: rows an odd number of 4 KiB pages   profiles/r04/stride_probe4-6.log    yes
 *   tally_window   1        0, 1       column-view tallies summed per workgroup in LDS around its photons' tile       profiles/r04/ab_no_tally_ablation   yes
 *   tally_lists    1        0, 1       flux tallies as sorted records (0: an atomic per level crossing)               profiles/r03/flux_tally_routes.log  yes
 *   tally_runs     1        0, 1       ... a flight through uniform layers as ONE run record, expanded by k_tl_runs   profiles/r06/ab_flux_run_records    yes
 *   entry_records  1        0, 1       new photons from a pre-pass kernel (48 bytes each, never > 1/2 of free memory)  profiles/r04/ab_block_c_entry_*     yes
 *   cam_images     -1       -1..8      cameras: periodic images of the camera served within this many domain lengths  profiles/r04/camera_images.log      no
 *                                      (-1: 2 where the ray kernel serves the job, else the nearest with a warning)
 *   vpad_col/_row  0        0..4096    unused 16-byte records after every column / row of the voxel records           profiles/r04/stride_probe*.log      yes
 *   overlap_rays   0        0, 1       two sets of event lists, ray kernels beside the next photon loop (-2.4 %: off)  profiles/r05/ab_overlap_rays.log    yes
 *   overlap_sort   1        0, 1, 2    flux jobs: two sets of record lists, the sort of launch i on a stream of its   profiles/r06/ab_flux_schedule.log   yes
 *                                      own beside the photon loop of launch i + 1; 0: one stream; 2: also across runs
 *                                      (see mi3d_run: the one case in which a run returns before its last sort joins)
 *   overlap_pre    1        0, 1, 2    photon order / entry records of launch i + 1 beside the photon loop of          profiles/r05/ab_overlap_pre.log     yes
 *                                      launch i (flux loop, event-writing loop); 2: also for small runs read one by one
 *   tl_split       4        1..64      with overlap_sort a run is worked off in at least this many launches            profiles/r05/ab_flux_sort_overlap   yes
 *   rays_wg        0        0..8       workgroups per CU of the ray kernel's light build (0: its own figure, 6)        profiles/r05/ab_knobs_after_nt.log  yes
 *   emit_wg        0        0..8       ... of the event-writing photon loop (0: 5)                                     profiles/r05/ab_knobs_after_nt.log  yes
 *   own_stream     0        0, 1       a non-blocking stream of the handle's own instead of the NULL stream            tools/time_dropin.py                no
 */
int mi3d_set_tuning(mi3d_solver *h, const char *key, int value);

/* Milliseconds spent in transport kernels since the last reset (HIP events on the launch
 * stream) and the number of launches.  Runs whose kernels use two streams (overlap_rays, overlap_sort) are timed as a whole, from their
 * first kernel to their last; under overlap_sort consecutive runs overlap (the next run's photon loop beside this run's last sort), so
 * the sum over runs can exceed the wall time. */
int mi3d_get_timing(mi3d_solver *h, double *kernel_ms, uint64_t *launches);

/* Normalised results.  `nphoton_total` is the number of histories the tallies hold (for a
 * photon-sharded job: the sum over all shards, after the all-reduce).
 *   radiance out[nview][nyr][nxr]   per unit Src_flx, i.e. the content of the reference's
 *                                    radiance out.bin (mca_out.py:467-481 then scales it)
 *   flux     out[3][nz+1][ny][nx]   direct-down, total-down, up */
int mi3d_get_radiance(mi3d_solver *h, uint64_t nphoton_total, float *out);
int mi3d_get_flux(mi3d_solver *h, uint64_t nphoton_total, float *out);
/* The part of the direct-down and total-down flux the kernels do not tally because it is known (the direct beam above the 3-D
 * region and in the horizontally uniform layers at its top: DESIGN.md §3): out[nz+1], per level, in the units of mi3d_get_flux
 * (Src_flx mu0 exp(-tau/mu0); 0 at the levels where every crossing is tallied), for the job that ran last on this handle.
 * mi3d_get_flux adds it by itself; a caller that normalises all-reduced RAW tallies of several jobs at once (one exchange per
 * batch of jobs instead of one per job) takes it from here, job by job. */
int mi3d_get_direct_levels(mi3d_solver *h, double *out);
/*   heating  out[nz][ny][nx]        absorbed radiant power per unit volume of every cell, per unit Src_flx [1/m x the unit of
 *                                    Src_flx]: (weight absorbed in the cell) x Src_flx mu0 nx ny / N / layer thickness.  The
 *                                    fourth variable ("hrt", nz layers) of the flux out.bin of a job with Flx_mhrt = 1; divided by
 *                                    (air density x c_p) it is the heating rate in K/s.  (The reference's reader has no branch for
 *                                    it, er3t/rtm/mca/mca_out.py:202-205; MCARaTS' own unit for the variable is not in the tree.) */
int mi3d_get_heating(mi3d_solver *h, uint64_t nphoton_total, float *out);
int mi3d_get_counters(mi3d_solver *h, uint64_t out[MI3D_NCOUNTER]);

/* ---- Run statistics on the device -------------------------------------------------------------
 * What the reference's reader does on the host with one file per (run, g) job
 * (er3t/rtm/mca/mca_out.py:313-352 flux, 438-500 radiance): every job's result is scaled by a
 * per-g factor (per level: the slit function varies with altitude) and summed over g; mean and
 * population standard deviation are then taken over the runs.  Here the normalised tallies of the
 * job that just ran are folded into a per-run field, and closed runs into float64 sums of x and
 * x*x per pixel, so that no per-job result has to leave the device.
 *
 *   mi3d_stats_begin    allocate and zero.  `rad_run` / `flux_run` are optional caller-owned
 *                       DEVICE buffers (float32, shapes of mi3d_get_radiance / mi3d_get_flux) for
 *                       the per-run fields: a photon-sharded job all-reduces them once per run
 *                       before mi3d_stats_end_run instead of once per job.
 *   mi3d_stats_add      run field += factor[level] * normalised tally of the current job (then
 *                       call mi3d_reset before the next job).  factor_rad[nview],
 *                       factor_flux[nz+1]; NULL = 1.  `nphoton_total` as in mi3d_get_radiance.
 *   mi3d_stats_end_run  close the run; optionally copy its field to the host (mode='all').
 *   mi3d_stats_get      mean and standard deviation over the closed runs; `which` is
 *                       MI3D_TARGET_RADIANCE or MI3D_TARGET_FLUX; any output may be NULL. */
int mi3d_stats_begin(mi3d_solver *h, void *rad_run, void *flux_run);
/* Two handles on one device sharing the jobs of a run (each transports every other job: the tail of one job's launch runs
 * beside the next job's).  mi3d_stats_join(h, owner): h adds its jobs (mi3d_stats_add) into the run fields of `owner`, which has
 * called mi3d_stats_begin; the run is closed and read on the owner.  mi3d_stats_chain(h, after): h's next statistics kernel
 * (mi3d_stats_add, mi3d_stats_end_run) waits for the last one of `after`; called before each of them, the run field is summed
 * in job order, as one handle sums it (er3t/rtm/mca/mca_out.py:313-352: the reader's g loop). */
int mi3d_stats_join(mi3d_solver *h, mi3d_solver *owner);
int mi3d_stats_chain(mi3d_solver *h, mi3d_solver *after);
/* Photon-sharded jobs (one process per GPU, each transporting a share of a job's photon ids and the run fields summed by
 * one all-reduce per run): the direct beam above the 3-D region is not tallied but known and joins the run
 * field in mi3d_stats_add -- on ONE rank only, or the sum over ranks would hold it world-size times.  share = 1 (default):
 * this handle adds the analytic term; share = 0: it does not (ranks > 0). */
int mi3d_stats_set_analytic_share(mi3d_solver *h, double share);
int mi3d_stats_add(mi3d_solver *h, uint64_t nphoton_total, const float *factor_rad, const float *factor_flux);
int mi3d_stats_end_run(mi3d_solver *h, float *rad_run_out, float *flux_run_out);
int mi3d_stats_get(mi3d_solver *h, int which, float *mean, float *sdev, int *nrun);

/* Test hook: fill out[4*n] with Philox4x32-10 words for counters (id0+i, draw) under `seed`,
 * computed on the device.  Lets the tests prove the device and oracle streams are bit-identical. */
int mi3d_debug_philox(mi3d_solver *h, uint64_t seed, uint64_t id0, uint32_t draw, int n,
                      uint32_t *out);
/* Test hook: the photon order of the LAST launch of the last mi3d_run (indices into the launch's id range sorted by start tile,
 * k_bin_*: a permutation of 0 .. n-1 for a launch of n photons) and, optionally, where each tile's piece of it ends (the cursors
 * k_bin_scatter leaves behind, which the lean loop's tally window reads; up to ntile_max words, 1024 at most).  MI3D_ESTATE when the
 * launch ran in id order (a small domain, fewer than 4096 photons, "tile_cols" 0). */
int mi3d_debug_order(mi3d_solver *h, uint64_t n, uint32_t *order_out, uint32_t *tile_end_out, int ntile_max);

#ifdef __cplusplus
}
#endif
#endif /* MI3D_H */
