"""Per-kernel totals out of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes
DIR/NAME_results.db where the CSV writer is not selected): the `top_kernels` view, printed as the table the *_kernel_stats.csv
files of profiles/ hold.  usage: python tools/rocpd_top_kernels.py gpurun_out/prof_tri2/tri_results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print('"Name","Calls","TotalDurationUs","AverageUs","Percentage"')
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print(f'"{name}",{calls},{total:.3f},{avg:.3f},{pct:.2f}')
