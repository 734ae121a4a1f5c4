#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) as CSV: name,calls,total_us,avg_us,pct.
usage: rocpd_summary.py results.db > profiles/<round>_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("kernel,calls,total_us,avg_us,pct,vgpr,sgpr,lds_bytes,grid,workgroup")
for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    row = db.execute("select vgpr_count,sgpr_count,lds_size,grid_x,grid_y,grid_z,workgroup_x from kernels where name=? limit 1",
                     (name,)).fetchone()
    short = name.split("(")[0].replace("void ", "")
    print('"%s",%d,%.1f,%.2f,%.2f,%s,%s,%s,%sx%sx%s,%s' % (short, calls, tot, avg, pct, row[0], row[1], row[2], row[3], row[4], row[5], row[6]))
