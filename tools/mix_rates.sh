#!/bin/bash
# what is the headline loop bound by?  Usage (GPU box): bash tools/mix_rates.sh <outdir> [workload] [photons]
#   1. tools/microbench/mix_rates ops                      -- every opcode class at 4 / 6 / 8 forced waves per SIMD, by wall time
#   2. rocprofv3 --pmc SQ_INSTS_VALU_* on one launch of the real kernel -> its DYNAMIC instruction mix by class
#   3. tools/microbench/mix_rates mix <that mix>           -- the ceiling of that mix: wave-instructions per second, all lanes, no memory
#   4. rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU on both: what the hardware's own busy counter
#      reads on a loop that only issues, and on the real kernel
O=${1:-gpurun_out/mix}; W=${2:-les480}; N=${3:-2e8}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
MB=$R/tools/microbench/mix_rates
[ -x $MB ] || hipcc --offload-arch=gfx950 -O3 $R/tools/microbench/mix_rates.hip -o $MB || exit 1
timeout -k 10 300 $MB ops > $R/$O/mix_rates_ops.log 2>&1 || exit 1
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"; do
  tag=$(echo $grp | awk '{print $2}')
  timeout -k 10 300 rocprofv3 --pmc $grp -d $R/$O/cls_$tag -o p --output-format csv -- python3 $R/tools/pmc_run.py $N $W > $R/$O/cls_$tag.log 2>&1 || exit 1
done
python3 - $R/$O $N > $R/$O/mix_args.txt <<'PY'
import csv, glob, sys, collections
d, n = sys.argv[1], float(sys.argv[2])
v = collections.Counter()
for f in glob.glob(d + '/cls_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_transport_lean' in row['Kernel_Name'] or 'k_transport_flux' in row['Kernel_Name'] or 'k_rays' in row['Kernel_Name']:
            v[row['Counter_Name']] += float(row['Counter_Value'])
tot = v['SQ_INSTS_VALU']
g = lambda k: v['SQ_INSTS_VALU_' + k]
known = g('ADD_F32') + g('MUL_F32') + g('FMA_F32') + g('TRANS_F32') + g('INT32') + g('INT64') + g('CVT')
other = max(tot - known, 0.0)
sys.stderr.write('per photon: total %.1f  add %.1f mul %.1f fma %.1f trans %.1f int32 %.1f int64 %.1f cvt %.1f other %.1f\n' %
                 tuple(x/n for x in (tot, g('ADD_F32'), g('MUL_F32'), g('FMA_F32'), g('TRANS_F32'), g('INT32'), g('INT64'), g('CVT'), other)))
# classes of the mix loop: int32 splits into xor-like and v_mul_lo_u32 (static listing: 19 of ~330 integer instructions);
# what the class counters do not name is compares, selects and moves (static listing: 2 : 2 : 3)
print('fma=%.1f mul=%.1f add=%.1f trans=%.1f int=%.1f mullo=%.1f mad64=%.1f cmp=%.1f cnd=%.1f pk=0 mov=%.1f' %
      (g('FMA_F32')/n, g('MUL_F32')/n, g('ADD_F32')/n, g('TRANS_F32')/n, 0.94*g('INT32')/n, 0.06*g('INT32')/n, g('INT64')/n,
       other*2/7/n, other*2/7/n, (other*3/7 + g('CVT'))/n))
PY
cat $R/$O/mix_args.txt
timeout -k 10 300 $MB mix $(cat $R/$O/mix_args.txt) > $R/$O/mix_rates_mix.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $R/$O/busy_mix -o p --output-format csv -- $MB mix $(cat $R/$O/mix_args.txt) > $R/$O/busy_mix.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $R/$O/busy_real -o p --output-format csv -- python3 $R/tools/pmc_run.py $N $W > $R/$O/busy_real.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/$O/wait_real -o p --output-format csv -- python3 $R/tools/pmc_run.py $N $W > $R/$O/wait_real.log 2>&1 || exit 1
python3 - $R/$O <<'PY' > $R/$O/busy_summary.txt
import csv, glob, sys, collections
d = sys.argv[1]
for tag in ('busy_mix', 'busy_real', 'wait_real'):
    rows = collections.defaultdict(lambda: collections.Counter()); disp = collections.Counter()
    for f in glob.glob(d + '/' + tag + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].split('(')[0][-60:]
            rows[k][row['Counter_Name']] += float(row['Counter_Value']); disp[k] += 1
    for k, c in rows.items():
        if c.get('SQ_INSTS_VALU', 1) < 1e6 and tag != 'wait_real':
            continue
        line = '%-10s %-62s' % (tag, k) + ' '.join('%s=%.4g' % kv for kv in sorted(c.items()))
        if 'SQ_ACTIVE_INST_VALU' in c and c.get('SQ_BUSY_CYCLES'):
            line += '  | ACTIVE_INST_VALU*4/BUSY_CYCLES=%.3f  ACTIVE_INST_VALU/INSTS_VALU=%.3f' % (4*c['SQ_ACTIVE_INST_VALU']/c['SQ_BUSY_CYCLES'], c['SQ_ACTIVE_INST_VALU']/max(c['SQ_INSTS_VALU'], 1))
        if 'SQ_WAVE_CYCLES' in c:
            line += '  | WAIT_ANY/WAVE_CYCLES=%.3f WAIT_INST_ANY/WAVE_CYCLES=%.3f ACTIVE_INST_ANY/WAVE_CYCLES=%.3f' % tuple(c[q]/c['SQ_WAVE_CYCLES'] for q in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'))
        print(line)
PY
cat $R/$O/busy_summary.txt
