"""tools/profile_summary.py -- condenses a rocprofv3 `*_kernel_stats.csv` into a short table that can be
committed under profiles/ (kernel names shortened, top-N by total time, plus totals)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"<.*", "<...>", name) if len(name) > 90 else name
    return name[:110]


def main(path, top=45):
    rows = list(csv.DictReader(open(path)))
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
    print("kernel,calls,total_ms,avg_us,pct")
    for r in rows[:top]:
        print('"%s",%s,%.3f,%.1f,%.2f' % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                          float(r["AverageNs"]) / 1e3, 100.0 * int(r["TotalDurationNs"]) / total))
    rest = rows[top:]
    print('"(%d other kernels)",%d,%.3f,,%.2f' % (len(rest), sum(int(r["Calls"]) for r in rest),
                                                  sum(int(r["TotalDurationNs"]) for r in rest) / 1e6,
                                                  100.0 * sum(int(r["TotalDurationNs"]) for r in rest) / total))
    print('"TOTAL",%d,%.3f,,100' % (sum(int(r["Calls"]) for r in rows), total / 1e6))


def from_trace(path, marker, skip, top=45):
    """stats of the kernel TRACE restricted to launches after the `skip`-th launch of a kernel whose name
    contains `marker` (drops warm-up, e.g. MIOpen's solver-search trials)"""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seen, t0 = 0, None
    for r in rows:
        if marker in r["Kernel_Name"]:
            seen += 1
            if seen == skip + 1:
                t0 = int(r["Start_Timestamp"])
                break
    agg = {}
    for r in rows:
        if t0 is not None and int(r["Start_Timestamp"]) < t0:
            continue
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(r["Kernel_Name"], [0, 0])
        a[0] += 1
        a[1] += d
    total = sum(v[1] for v in agg.values())
    span = int(rows[-1]["End_Timestamp"]) - (t0 or int(rows[0]["Start_Timestamp"]))
    # union of the kernel intervals = time during which at least one kernel was resident; the rest is device idle
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if t0 is None or int(r["Start_Timestamp"]) >= t0)
    union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    union += cur_e - cur_s
    print("# kernels after the %d-th launch of *%s*: busy %.1f ms of a %.1f ms window" % (skip, marker, total / 1e6, span / 1e6))
    print("# at least one kernel resident for %.1f ms (%.1f %% of the window); device idle %.1f ms" % (
        union / 1e6, 100.0 * union / span, (span - union) / 1e6))
    print("kernel,calls,total_ms,avg_us,pct")
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    for k, (c, t) in items[:top]:
        print('"%s",%d,%.3f,%.1f,%.2f' % (short(k), c, t / 1e6, t / c / 1e3, 100.0 * t / total))
    rest = items[top:]
    print('"(%d other kernels)",%d,%.3f,,%.2f' % (len(rest), sum(v[0] for _, v in rest), sum(v[1] for _, v in rest) / 1e6,
                                                  100.0 * sum(v[1] for _, v in rest) / total))


def gaps(path, marker, skip, top=25):
    """largest device-idle gaps of the window, with the kernels before and after each"""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seen, t0 = 0, int(rows[0]["Start_Timestamp"])
    for r in rows:
        if marker in r["Kernel_Name"]:
            seen += 1
            if seen == skip + 1:
                t0 = int(r["Start_Timestamp"])
                break
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
    out, cur_e, last = [], int(rows[0]["End_Timestamp"]), rows[0]
    hist = {}
    for r in rows[1:]:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if a > cur_e:
            out.append((a - cur_e, (cur_e - t0) / 1e6, short(last["Kernel_Name"])[:60], short(r["Kernel_Name"])[:60]))
            bucket = min(int((a - cur_e) / 1e3) // 10 * 10, 200)
            hist[bucket] = hist.get(bucket, 0) + (a - cur_e)
        if b > cur_e:
            cur_e, last = b, r
    print("# idle by gap length (us bucket -> total ms): " + ", ".join("%d+: %.2f" % (k, v / 1e6) for k, v in sorted(hist.items())))
    for g, at, before, after in sorted(out, reverse=True)[:top]:
        print("%8.1f us idle at %8.2f ms   after %-60s before %s" % (g / 1e3, at, before, after))


def family(name):
    n = name
    for key, fam in (("vit_gemm", "vit_gemm"), ("vit_attention", "attention"), ("conv_igemm", "conv"), ("conv_wgrad", "wgrad"), ("wgrad_fold", "wgrad"),
                     ("igemm_", "miopen"), ("grouped_conv", "miopen"), ("raster_", "raster"), ("fvm_", "corr"), ("cols_partial", "corr"),
                     ("dual_backward", "corr"), ("Cijk_", "blas"), ("bn_", "bn"), ("bias_leaky", "bn"), ("multi_tensor", "optim"),
                     ("row_stats", "vit_small"), ("mutual_nn", "corr"), ("nearest", "sym"), ("upsample", "up"), ("maxpool", "pool")):
        if key in n:
            return fam
    return "other"


def timeline(path, marker, skip, bin_us=500.0):
    """one training step (between the skip-th and the (skip+1)-th launch of `marker`) in bins: for every bin, per kernel family, the
    time-averaged number of resident kernels (1.0 = one kernel of that family resident during the whole bin), per queue"""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if marker in r["Kernel_Name"]]
    # the optimizer launches several multi_tensor kernels per step: a step boundary = a gap of > 5 ms between marker launches
    bounds = [marks[0]] + [b for a, b in zip(marks, marks[1:]) if b - a > 5e6]
    t0, t1 = bounds[skip], bounds[skip + 1]
    nb = int((t1 - t0) / (bin_us * 1e3)) + 1
    fams, queues = {}, {}
    for r in rows:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if b <= t0 or a >= t1:
            continue
        f = family(r["Kernel_Name"])
        q = r.get("Queue_Id", "?")
        for tab, key in ((fams, f), (queues, q)):
            arr = tab.setdefault(key, [0.0] * nb)
            i0, i1 = max(int((a - t0) / (bin_us * 1e3)), 0), min(int((b - t0) / (bin_us * 1e3)), nb - 1)
            for i in range(i0, i1 + 1):
                lo, hi = t0 + i * bin_us * 1e3, t0 + (i + 1) * bin_us * 1e3
                arr[i] += max(0.0, min(b, hi) - max(a, lo)) / (bin_us * 1e3)
    order = sorted(fams, key=lambda k: -sum(fams[k]))
    print("# step of %.2f ms, %.0f us bins; residency per kernel family (then per queue)" % ((t1 - t0) / 1e6, bin_us))
    print("t_ms  " + " ".join("%9s" % k[:9] for k in order) + " | " + " ".join("q%-4s" % str(k)[-4:] for k in sorted(queues)))
    for i in range(nb):
        print("%5.1f " % (i * bin_us / 1e3) + " ".join("%9.2f" % fams[k][i] if fams[k][i] >= 0.005 else "        ." for k in order) + " | "
              + " ".join("%5.2f" % queues[k][i] for k in sorted(queues)))


def chain_census(path, marker, skip, start_key="fvm_forward", end_key="bn_bwd", top=40):
    """the step's serial middle part (from the first `start_key` launch of a step to the first `end_key` launch after it): kernels
    aggregated by name, device-idle time and how much of the window had exactly one kernel resident"""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if marker in r["Kernel_Name"]]
    bounds = [marks[0]] + [b for a, b in zip(marks, marks[1:]) if b - a > 5e6]
    t0, t1 = bounds[skip], bounds[skip + 1]
    step = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
    a = next(int(r["Start_Timestamp"]) for r in step if start_key in r["Kernel_Name"])
    b = next(int(r["Start_Timestamp"]) for r in step if end_key in r["Kernel_Name"] and int(r["Start_Timestamp"]) > a)
    win = [r for r in step if int(r["End_Timestamp"]) > a and int(r["Start_Timestamp"]) < b]
    agg = {}
    for r in win:
        d = min(int(r["End_Timestamp"]), b) - max(int(r["Start_Timestamp"]), a)
        e = agg.setdefault(short(r["Kernel_Name"])[:70], [0, 0])
        e[0] += 1
        e[1] += d
    # coverage profile
    ev = []
    for r in win:
        ev.append((max(int(r["Start_Timestamp"]), a), 1))
        ev.append((min(int(r["End_Timestamp"]), b), -1))
    ev.sort()
    depth, last, hist = 0, a, {}
    for t, d in ev:
        hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
        depth += d
        last = t
    print("# serial middle part of one step: %.2f ms (%s -> %s), %d launches" % ((b - a) / 1e6, start_key, end_key, len(win)))
    print("# resident kernels 0 / 1 / 2 / >=3: " + " / ".join("%.2f ms" % (hist.get(k, 0) / 1e6) for k in range(4)))
    print("kernel,calls,total_us")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('"%s",%d,%.1f' % (k, c, t / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "--chain":
        chain_census(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    elif sys.argv[1] == "--timeline":
        timeline(sys.argv[2], sys.argv[3], int(sys.argv[4]), float(sys.argv[5]) if len(sys.argv) > 5 else 500.0)
    elif sys.argv[1] == "--gaps":
        gaps(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 25)
    elif sys.argv[1] == "--trace":
        from_trace(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 45)
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45)
