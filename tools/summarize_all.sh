#!/bin/bash
# gpurun_out/<tag>_* (tools/profile_bench.sh) -> profiles/<tag>_*
set -e
tag=${1:-prof}
g=gpurun_out
f() { find $g/${tag}_$1 -name "$2" | head -1; }
python tools/summarize_profile.py stats "$(f stats 's_kernel_stats.csv')" profiles/${tag}_kernel_stats.csv
python tools/summarize_profile.py traffic "$(f fetch 'f_counter_collection.csv')" "$(f write 'w_counter_collection.csv')" profiles/${tag}_hbm_traffic.json
python tools/summarize_profile.py stats "$(f score 's_kernel_stats.csv')" profiles/${tag}_scoring_kernel_stats.csv
python tools/summarize_profile.py traffic "$(f sfetch 'f_counter_collection.csv')" "$(f swrite 'w_counter_collection.csv')" profiles/${tag}_scoring_hbm_traffic.json
