#!/bin/bash
# A/B of tuning knobs that are environment variables of ONE build: tools/ab_env.sh <photons> <workload> "VAR=val VAR2=val" "..." ...
N=$1; W=$2; shift 2
for spec in "$@"; do
  echo "== $spec"
  env $spec AB_WORKLOAD=$W python tools/ab.py $N er3t_amd/libmi3drt.so
done
