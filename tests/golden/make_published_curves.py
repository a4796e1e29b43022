#!/usr/bin/env python3
"""Extract the one recorded output of the COMMITTED plot_errorVSsnr.m from the reference tree into a small data fixture.

    python tests/golden/make_published_curves.py     ->  tests/golden/errorVSsnr_angles_published.json

/root/reference/results/errorVSsnr_angles.fig is a MAT v7.3 (HDF5) figure saved by plot_errorVSsnr.m:211.  There is no
h5py in the image, so the numbers are found by scanning the file for the float64 XData pattern (-15:3:15, :24) and reading
the 11-double YData block stored next to each occurrence; series are labelled by the order of the legend strings
(VAMP, MMV-OMP, Proposed, Proposed with angle information).  Each value is ONE
unseeded realisation per SNR point (maxMCRealizations = 1, :18) - a sample of the reference's output distribution,
not a repeatable number.  The fixture holds data only (no reference source text)."""
import json
import os
import struct

import numpy as np

SRC = "/root/reference/results/errorVSsnr_angles.fig"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "errorVSsnr_angles_published.json")


def main():
    raw = open(SRC, "rb").read()
    x = np.arange(-15, 16, 3, dtype="<f8")                          # plot_errorVSsnr.m:24
    pat = x.tobytes()
    hits, i = [], raw.find(pat)
    while i >= 0:
        hits.append(i)
        i = raw.find(pat, i + 1)
    # the line objects are stored twice (object tree + a second serialisation); YData sits 360 bytes after XData
    # (the first XData block belongs to a line without stored YData; the four that follow carry the DisplayName strings
    #  "VAMP", "MMV-OMP", "Proposed", "Proposed with angle information", in this order, right before their data)
    names = ["VAMP", "MMV-OMP", "Proposed", "Proposed with angle information"]
    out = {}
    for name, h in zip(names, hits[1:5]):
        y = np.frombuffer(raw[h + 360:h + 360 + 88], dtype="<f8")
        assert np.all(np.isfinite(y)) and np.all(y > 0) and np.all(y <= 1.0 + 1e-12), (name, y)
        out[name] = [float(v) for v in y]
    for name, h in zip(["Proposed with angle information", "Proposed", "MMV-OMP"], hits[5:8]):   # second copy agrees
        y = np.frombuffer(raw[h + 360:h + 360 + 88], dtype="<f8")
        assert np.array_equal(y, np.array(out[name])), name
    fixture = {
        "provenance": "results/errorVSsnr_angles.fig of vlaxose/jstsp19 (saved by plot_errorVSsnr.m:211; MAT v7.3, YData read "
                      "360 bytes after each float64 XData block -15:3:15); ONE unseeded realisation per point "
                      "(maxMCRealizations = 1, :18), parameters of plot_errorVSsnr.m:8-25 (Nr=32, Nt=4, L=4, Mr=4, T=35, "
                      "Imax=100, numOfnz=100); capped spectral NMSE (:138-141)",
        "snr_db": [float(v) for v in x],
        "series": out,
    }
    with open(OUT, "w") as f:
        json.dump(fixture, f, indent=1)
    for k, v in out.items():
        print("%-34s" % k, " ".join("%.4g" % t for t in v))


if __name__ == "__main__":
    main()
