"""Turn a rocprofv3 rocpd database (…_results.db) into the text summary committed under profiles/."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print(f"{'calls':>7} {'total_us':>14} {'avg_us':>12} {'%':>7}  kernel")
for name, calls, tot, avg, pct in rows:
    short = name.split("(")[0]
    if "rocprim" in short:
        short = "rocprim::" + ("radix_sort_onesweep_iteration" if "onesweep_iteration" in name else "radix_sort_histogram" if "global_offsets" in name else short[:60])
    print(f"{calls:>7} {tot:>14.3f} {avg:>12.3f} {pct:>7.3f}  {short}")
