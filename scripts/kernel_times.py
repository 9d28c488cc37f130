"""Dev helper: per-kernel average durations from a rocprofv3 rocpd database (results.db)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select name, total_calls, average from top_kernels"):
    print(f"{r[0][:70]:70s} calls {r[1]:5d}  avg {r[2] / 1e3:9.3f} ms")
