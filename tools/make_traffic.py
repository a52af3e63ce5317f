"""profiles/traffic.json from the two PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench command.
   usage: tools/make_traffic.py <fetch_dir> <write_dir> <workload> <photons_per_launch> <out.json> [<sq_insts_valu_dir>]
   gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; FETCH_SIZE reads half of
   the bytes of wide coalesced streams (factor 2 applied, recorded separately: this kernel's gathers are 4-byte
   random reads, for which the guide calls the factor uncalibrated); WRITE_SIZE is exact for float atomics."""
import csv, glob, json, sys
fd, wd, work, nph, out = sys.argv[1:6]
def avg(d, name):
    vals = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'k_transportILb0' in row['Kernel_Name'] or ('k_transport<false' in row['Kernel_Name']):
                if row['Counter_Name'] == name:
                    vals.append(float(row['Counter_Value']))
    return sum(vals)/len(vals), len(vals)
fetch, nf = avg(fd, 'FETCH_SIZE'); write, nw = avg(wd, 'WRITE_SIZE')
rec = {'fetch_size_kib_per_launch': fetch, 'write_size_kib_per_launch': write, 'launches_averaged': [nf, nw],
       'hbm_bytes_per_launch_uncorrected': (fetch+write)*1024.0,
       'hbm_bytes_per_launch': (2.0*fetch+write)*1024.0,
       'note': 'hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the guide; Infinity-Cache hits are included in these fabric counters'}
if len(sys.argv) > 6:
    valu, nv = avg(sys.argv[6], 'SQ_INSTS_VALU')
    rec['valu_insts_per_launch'] = valu
    rec['launches_averaged'].append(nv)
try:
    tj = json.load(open(out))
except Exception:
    tj = {}
tj['%s:%d' % (work, int(float(nph)))] = rec
json.dump(tj, open(out, 'w'), indent=1)
print(json.dumps(rec))
