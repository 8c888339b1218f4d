#!/bin/bash
# tools/pmc_kernels.sh <kernel-name regex> <counters, <= 8 SQ per pass> -- <command...>
# One rocprofv3 --pmc pass (kernel-trace only, as the pool requires); per matching kernel: launches and the mean per launch of each
# counter.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md).
pat="$1"; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
export TMPDIR=/tmp
out=$(mktemp -d /tmp/pmc.XXXXXX)
( cd /tmp && rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d "$out" -- "$@" > "$out/run.log" 2>&1 )
f=$(ls "$out"/*/*counter_collection.csv 2>/dev/null | head -1)
[ -z "$f" ] && { tail -5 "$out/run.log"; exit 1; }
python3 - "$f" "$pat" <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
pat = re.compile(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if pat.search(r["Kernel_Name"]):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print("== " + k[:110])
    for c, v in sorted(agg[k].items()):
        print("   %-30s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
