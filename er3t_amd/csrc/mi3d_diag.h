// mi3d_diag.h -- everything the photon loops carry for MEASUREMENT builds only, in one place, so that the kernels read as the algorithm
// (VERDICT r5 item 6).  None of it is compiled into the shipped library: every macro below expands to nothing unless the build asks
// for it (make EXTRA=-DMI3D_...).  What each build is for, and the log it produced, stands beside its switch.
#pragma once

// -DMI3D_MARKS: comments in the ISA listing that delimit the blocks of a loop (tools/isa_blocks.py; superseded by the line table,
// tools/isa_lines.py, but still the quickest way to find a block in a listing)
#ifdef MI3D_MARKS
#define MI3D_MARK(name) asm volatile("; MARK " name)
#else
#define MI3D_MARK(name)
#endif

// The instrumented builds (COUNT = true, mi3d_set_counting) split the wave time of a loop over six clock counters (Counters::cyc).
// -DMI3D_NO_LEAN_TICKS leaves the counters to the ray kernels, which share them (tools/sched_rays.py); -DMI3D_CENSUS keeps three
// (the census of tally addresses uses the others: tools/census.py, profiles/r04/tally_same_address_census.log).
#if defined(MI3D_NO_LEAN_TICKS)
#define MI3D_DIAG_TICK(COUNT_, cnt_, tick_, slot) do { } while (0)
#elif defined(MI3D_CENSUS)
#define MI3D_DIAG_TICK(COUNT_, cnt_, tick_, slot) do { if ((COUNT_) && (slot) < 3) { const long long t_ = clock64(); (cnt_).cyc[slot] += (uint32_t)((t_ - (tick_)) >> 6); (tick_) = t_; } } while (0)
#else
#define MI3D_DIAG_TICK(COUNT_, cnt_, tick_, slot) do { if (COUNT_) { const long long t_ = clock64(); (cnt_).cyc[slot] += (uint32_t)((t_ - (tick_)) >> 6); (tick_) = t_; } } while (0)
#endif

// -DMI3D_WIN_DIAG: le_steps3d counts the column-view tallies that stayed in the LDS tally window, le_steps all of them (82-90 %,
// profiles/r04/win_offset_probe.log)
#ifdef MI3D_WIN_DIAG
#define MI3D_WIN_HIT(COUNT_, cnt_) do { if (COUNT_) (cnt_).le_steps3d++; } while (0)
#define MI3D_WIN_ANY(COUNT_, cnt_) do { if (COUNT_) (cnt_).le_steps++; } while (0)
#else
#define MI3D_WIN_HIT(COUNT_, cnt_) do { } while (0)
#define MI3D_WIN_ANY(COUNT_, cnt_) do { } while (0)
#endif

// -DMI3D_CLEAR_STEPS: voxel steps through cloud-free voxels counted in le_steps, those that end the walk there in le_steps3d (12.1 of the
// 61.5 steps per photon on the bench scene: an empty-space structure would not pay)
#ifdef MI3D_CLEAR_STEPS
#define MI3D_DIAG_CLEAR_STEP(COUNT_, cnt_, ks_, ends_) do { if ((COUNT_) && (ks_) == 0.0f) { (cnt_).le_steps++; if (ends_) (cnt_).le_steps3d++; } } while (0)
#else
#define MI3D_DIAG_CLEAR_STEP(COUNT_, cnt_, ks_, ends_) do { } while (0)
#endif

// -DMI3D_ABL_NOEMITSTORE (results wrong): the event-writing loop without its stores -- what the event records cost it (21 %,
// profiles/r03, `kt_noemit`)
#ifdef MI3D_ABL_NOEMITSTORE
#define MI3D_DIAG_NOEMITSTORE 1
#define MI3D_DIAG_KEEP11(a, b, c, d, e, f, g, h, i, j, k) asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h), "v"(i), "v"(j), "v"(k))
#else
#define MI3D_DIAG_NOEMITSTORE 0
#endif
