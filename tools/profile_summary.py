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


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45)
