#!/bin/bash
# tools/pmc_kernel.sh <kernel-name-substring> <counter list, space separated, <= 8 SQ per pass> -- <command...>
# One rocprofv3 --pmc pass (kernel-trace only, as the pool requires); prints the mean per launch of each counter.
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
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-32s n=%-4d mean=%.4g" % (k, len(v), sum(v) / len(v)))
PY
