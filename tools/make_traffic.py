"""profiles/traffic.json from the PMC passes of ONE launch of the transport kernel (tools/pmc_mix.sh + the FETCH_SIZE and
WRITE_SIZE passes of tools/final_measure.sh).

    tools/make_traffic.py <pmc_dir> <workload> <photons_of_the_launch> <out.json> <session label>

Per-photon figures, keyed by workload (bench.py scales them to its photons per launch and says they are replayed).
Units and corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB.  FETCH_SIZE = 64 B x TCC_EA0_RDREQ.  The
guide doubles it for WIDE COALESCED streams (128-byte requests tallied at 64); this kernel's reads are 16-byte per-lane
gathers, for which profiles/r02/fetch_size_calibration_16B_gathers.txt measures 64 B per L2 miss at 5.5e10 misses/s from
HBM -- doubled, that would exceed the HBM peak, so the factor does not apply and is NOT applied here.  WRITE_SIZE is exact
for float atomics."""
import csv, glob, json, sys
pmc, work, nph, out, session = sys.argv[1:6]
nph = float(nph)
def val(name):
    """sum over every dispatch of the transport kernels (k_transport*, k_rays, k_tl_*) in the profiled process: tools/pmc_run.py makes
    exactly one run of <photons> histories, in one launch or several"""
    vals = []
    for f in glob.glob(pmc + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if any(k in row['Kernel_Name'] for k in ('k_transport', 'k_rays', 'k_tl_')) and row['Counter_Name'] == name:
                vals.append(float(row['Counter_Value']))
    return sum(vals) if vals else None
fetch, write = val('FETCH_SIZE'), val('WRITE_SIZE')
rec = {'session': session, 'photons_of_the_measured_run': nph}
if fetch is not None and write is not None:
    rec.update(fetch_bytes_per_photon=fetch*1024.0/nph, write_bytes_per_photon=write*1024.0/nph,
               hbm_bytes_per_photon=(fetch+write)*1024.0/nph,
               note='(FETCH_SIZE + WRITE_SIZE) x 1024; no x2 on FETCH_SIZE: 16-byte gathers, calibrated in profiles/r02/fetch_size_calibration_16B_gathers.txt')
hit, req = val('TCC_HIT_sum'), val('TCC_REQ_sum')
if hit is not None and req:
    rec['tcc_hit_rate'] = hit/req
atom = val('TCC_EA0_ATOMIC_sum')
if atom is not None:
    rec['l2_to_fabric_atomics_per_photon'] = atom/nph
valu, thr = val('SQ_INSTS_VALU'), val('SQ_THREAD_CYCLES_VALU')
if valu is not None:
    rec['valu_insts_per_photon'] = valu/nph
    if thr is not None:
        rec['lane_utilisation'] = thr/(64.0*valu)
try:
    tj = json.load(open(out))
except Exception:
    tj = {}
tj[work] = rec
json.dump(tj, open(out, 'w'), indent=1)
print(json.dumps(rec))
