#!/usr/bin/env python3
"""Extract the recorded convergence curves of the reference's results/errorVSadmmiters.fig into a small data fixture.

    python tests/golden/make_published_admmiters.py   ->  tests/golden/errorVSadmmiters_published.json

The file is a MAT-v5 figure container (scipy.io.loadmat reads it): four axes, each with two line series of 70 points -
the legend names them epsilon_1 and epsilon_2, the titles give (N_T, L_R, SNR) per panel.  It was saved by an OLDER
revision of plot_errorVSadmmiters.m than the committed one (70 iterations and two curves per panel instead of 100 and
four; titles 'N_T=4, L_R=24, SNR=5db' ... instead of 'N_T=4, T=10, SNR=15db' ...; values plotted raw although the axis
label says dB): the frame length, the number of delay taps and the realisation count behind it are not recorded.  The
fixture holds data only - the numbers and the strings stored in the figure."""
import json
import os

import numpy as np
import scipy.io as sio

SRC = "/root/reference/results/errorVSadmmiters.fig"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "errorVSadmmiters_published.json")


def main():
    m = sio.loadmat(SRC, squeeze_me=True, struct_as_record=False)
    fig = m["hgS_070000"]
    panels = []
    for ax in fig.children:
        if ax.type != "axes":
            continue
        title = [c.properties.String for c in ax.children if c.type == "text"][-1]
        lines = [c for c in ax.children if c.type == "graph2d.lineseries"]
        assert len(lines) == 2
        x = np.asarray(lines[0].properties.XData, dtype=float)
        assert np.array_equal(x, np.arange(1, 71))
        e1, e2 = (np.asarray(l.properties.YData, dtype=float) for l in lines)
        assert np.all(e1 > 0) and np.all(e2 > 0)
        panels.append({"title": str(title), "epsilon_1": [float(v) for v in e1], "epsilon_2": [float(v) for v in e2]})
    assert len(panels) == 4
    fixture = {
        "provenance": "results/errorVSadmmiters.fig of vlaxose/jstsp19 (MAT v5 figure, read with scipy.io.loadmat): four panels, "
                      "two line series each (legend: epsilon_1, epsilon_2), 70 iterations; saved by an older revision of "
                      "plot_errorVSadmmiters.m than the committed one - frame length, delay taps and realisation count unrecorded",
        "iterations": 70,
        "panels": panels,
    }
    with open(OUT, "w") as f:
        json.dump(fixture, f, indent=1)
    for p in panels:
        print(p["title"], "eps1 %.3g -> %.3g" % (p["epsilon_1"][0], p["epsilon_1"][-1]),
              "eps2 %.3g -> %.3g" % (p["epsilon_2"][0], p["epsilon_2"][-1]))


if __name__ == "__main__":
    main()
