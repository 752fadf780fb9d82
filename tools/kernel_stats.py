"""Per-kernel summary (calls, total, average, share) of a `rocprofv3 --kernel-trace --stats`
result database, as a markdown table:  python tools/kernel_stats.py t_results.db"""
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    print("| kernel | calls | total ms | average ms | % of GPU time |")
    print("|---|---|---|---|---|")
    for name, calls, total, avg, pct in cur.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels"):
        print("| `%s` | %d | %.3f | %.3f | %.3f |" % (name.replace("(anonymous namespace)::", "").split("(")[0], calls, total / 1e3, avg / 1e3, pct))


if __name__ == "__main__":
    main()
