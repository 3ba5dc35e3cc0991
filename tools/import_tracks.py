"""Re-encodes the reference's track centre lines (data/<track>/center_line.csv)
as the build's own input fixtures ihm2_amd/data/<track>.csv (full precision, uniform header).
Run in the build container only (reads /root/reference).  Data files, not source.
"""
import os
import numpy as np

SRC = "/root/reference/data"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ihm2_amd", "data")
os.makedirs(DST, exist_ok=True)
for track in sorted(os.listdir(SRC)):
    f = os.path.join(SRC, track, "center_line.csv")
    if not os.path.exists(f):
        continue
    arr = np.loadtxt(f, delimiter=",", skiprows=1)  # quirk Q10: short_skidpad has a '# x,y,..' header
    assert arr.ndim == 2 and arr.shape[1] == 4, (track, arr.shape)
    np.savetxt(os.path.join(DST, track + ".csv"), arr, delimiter=",", header="x,y,right_width,left_width",
               comments="", fmt="%.17g")
    print(track, arr.shape)
