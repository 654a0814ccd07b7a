"""conv_update under launch-geometry overrides (one process each: ULTRA_CONV_GEOMETRY is read once), at the headline size
(116,328 rows, cache resident) and an HBM-bound one (CoDEx-L bs 8: 623,608 rows).  argv: [lib variant path]"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
geos = sys.argv[1].split(";") if len(sys.argv) > 1 else ["", "512,256", "256,512", "256,768"]
for rows in (14541 * 8, 77951 * 8):
    for geo in geos:
        env = dict(os.environ)
        if geo:
            env["ULTRA_CONV_GEOMETRY"] = geo
        out = subprocess.run([sys.executable, os.path.join(here, "conv_probe.py"), str(rows)], env=env, capture_output=True, text=True)
        lines = [l for l in out.stdout.splitlines() if " us" in l and "floor" not in l]
        print("rows %7d geometry %-9s | full %s | no LN %s | no matrix %s | neither %s" %
              tuple([rows, geo or "default"] + [l.split()[-2] for l in lines[:4]]), flush=True)
        if len(lines) < 4:
            print(out.stdout[-500:], out.stderr[-1500:])
