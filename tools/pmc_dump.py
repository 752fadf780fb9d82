"""Per-dispatch averages of every counter in rocprofv3 PMC result databases, for kernels whose
name contains a substring:  python tools/pmc_dump.py <substr> a_results.db [b_results.db ...]"""
import sqlite3
import sys


def main():
    sub = sys.argv[1]
    for db in sys.argv[2:]:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) "
                           "from counters_collection group by kernel_name, counter_name")
        for name, counter, total, ndisp in rows:
            if sub in name:
                print("%-28s %-14.6g dispatches=%d  %s" % (counter, total / max(ndisp, 1), ndisp,
                                                          name.replace("(anonymous namespace)::", "").split("(")[0][:60]))


if __name__ == "__main__":
    main()
