#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/c30; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "les128_mie" > $O/pytest_mie.log 2>&1; echo "pytest mie rc $?"; grep "les128_mie\|passed\|failed" $O/pytest_mie.log | tail -12
timeout -k 10 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - $O/bench.log <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']
        print('headline %.4g photons/s ms/step %.2f frac %.3f frac_step %.3f sclk %s' % (d['value'], d['ms_per_step'], r['frac'], r['frac_step'], r['sclk_mhz']))
        for k,v in d.get('secondary',{}).items():
            if 'error' in v: print(' ', k, v['error']); continue
            p=v.get('parity') or {}
            print('  %-26s %.4g photons/s frac %.3f frac_step %.3f  parity sigma %s paired_rel %s paired_se %s' % (k, v['value'], v['roofline']['frac'], v['roofline']['frac_step'], p.get('domain_mean_diff_sigma'), p.get('paired_rel_diff'), p.get('paired_diff_in_paired_se')))
        print('  published_case', {k: d['published_case'].get(k) for k in ('seconds','seconds_mcarats_ng','seconds_mca_out_ng','kernel','error','rad_std_over_runs_rel')})
PY
