#!/bin/bash
# Wavefront batch solver on the GPU box: parity + kernel time against the generic kernel (tools/wave_check.py), then the
# cycle profile of its phases from the -DDNLP_WAVE_PROF build (dnlp_amd/libdnlp_hip_prof.so; built by
#   hipcc <flags of __graft_entry__.py> -DDNLP_WAVE_PROF capi.hip -o ../libdnlp_hip_prof.so).
# usage: tools/wave_profile.sh [templates] [batch]
cd "${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}"
W=${1:-localization}
B=${2:-8192}
timeout 600 python tools/wave_check.py --which $W --batch $B --reps 3 2>&1 | tail -4
if [ -f dnlp_amd/libdnlp_hip_prof.so ]; then
  DNLP_HIP_LIB=$PWD/dnlp_amd/libdnlp_hip_prof.so timeout 300 python tools/wave_check.py --which $W --batch $B --reps 1 --skip-generic --out prof.jsonl 2>&1 | grep -E "wave profile"
fi
