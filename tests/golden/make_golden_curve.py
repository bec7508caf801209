"""Reads the marker positions out of a reference-PRODUCED plot: README.assets/load-deflection-curve.png (README.md:95,
Fig. 2 (d): "Curves of vertical deflection vs. load" of the cantilever of tests/beam_deflection/load800_freeEnd_*:
FEMcy small deformation, FEMcy large deformation, Abaqus large deformation, eleven points each at loads 0, 80 ... 800 MPa)
and writes them as numbers to tests/golden/readme_load_deflection.json.  Runs in the build container only (needs
/root/reference and PIL); the picture itself is not copied, only the measured marker centres.

How: the plot is a matplotlib figure with default colours.  Axis calibration from the tick marks (outside the spines);
a marker is a filled disc of one colour, later series drawn over earlier ones (blue, orange, green), so an orange disc
may be partly covered by the green line or a green disc: its centre is found by sliding a disc of the markers' radius
over the neighbourhood of the known abscissa and counting own-colour pixels inside minus background pixels inside
(covered pixels count for nothing), at quarter-pixel steps.

Beside the readings the file gets the ORACLE's curve of the same decks (tests/golden/decks/
beamDeflec_quadPSE_{smallD_load800_freeEnd,largeD_load800}.inp = the reference's tests/beam_deflection/load800_freeEnd_*),
ten increments of 0.1, u_y of the node at the middle of the free end (40, 2): what the device driver is held to.

A second reference-produced record of the same cantilever: tests/beam_deflection/load800_freeEnd_largeDef/
beamDeflec_quadPSE_largeD_load800_stable.gif -- 21 frames = the window after every converged increment of a run in twenty
increments of 0.05 (stiffnessMtrx.py:668-711 shows / saves one picture at the start and one per increment; body.py:100-162
keeps ONE camera, looking straight at the plane of a 2-D body, so all frames share one scale).  Frame 0 is the undeformed
40 x 4 beam = 316 x 32 pixels: 7.9 pixels per unit.  Recorded: the bounding box of the non-black pixels of every frame.

usage: python tests/golden/make_golden_curve.py"""
import json
import os
import sys

import numpy as np
from PIL import Image
from scipy import ndimage

SRC = "/root/reference/README.assets/load-deflection-curve.png"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "readme_load_deflection.json")
COLOURS = {"small_deformation": (31, 119, 180), "large_deformation": (255, 127, 14)}
ABAQUS = (44, 160, 44)          # the third series (drawn last, never covered; its loads are not multiples of 80)


GIF = "/root/reference/tests/beam_deflection/load800_freeEnd_largeDef/beamDeflec_quadPSE_largeD_load800_stable.gif"


def gif_boxes():
    im = Image.open(GIF)
    boxes = []
    for i in range(im.n_frames):
        im.seek(i)
        a = np.asarray(im.convert("RGB")).astype(int)
        ys, xs = np.where(a.sum(axis=2) > 40)                       # black background
        boxes.append([int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)])
    return boxes


def oracle_boxes():
    """the deformed beam's bounding box (all nodes) after every increment of 0.05."""
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    from helpers import deck, oracle_system_from_inp
    from femcy_amd.reader import InpInfo
    inp = InpInfo(deck("beamDeflec_quadPSE_largeD_load800.inp"))
    s = oracle_system_from_inp(inp)
    boxes, newton = [[40.0, 4.0]], [0]
    advance = s.advance_inc

    def recording(bcs):
        ok, loops = advance(bcs)
        if ok:
            X = inp.nodes + s.dof.reshape(-1, 2)
            boxes.append([float(np.ptp(X[:, 0])), float(np.ptp(X[:, 1]))])
            newton.append(int(loops))
        return ok, loops
    s.advance_inc = recording
    s.solve(dict(inp.time_incs, ini_inc=0.05, max_inc=0.05), inp.dirichlet_bc_info, inp.neumann_bc_info)
    assert len(boxes) == 21 and all(i["converged"] for i in s.increments)
    return boxes, newton, s.n_solves


def oracle_curves():
    """u_y at the middle of the free end after every increment of 0.1: the linear deck from the undeformed state (one
    increment per load: the reference assembles K on nodes + dof, a second increment of a 'small deformation' run is
    not linear any more -- stiffnessMtrx.py:161-186) and the nlgeom deck by its Newton driver."""
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    from helpers import deck, oracle_system_from_inp
    from femcy_amd.reader import InpInfo
    inp = InpInfo(deck("beamDeflec_quadPSE_smallD_load800_freeEnd.inp"))
    tip = int(np.argmin(np.linalg.norm(inp.nodes - np.array([40.0, 2.0]), axis=1)))
    s = oracle_system_from_inp(inp)
    s.solve(dict(inp.time_incs, ini_inc=1.0, max_inc=1.0), inp.dirichlet_bc_info, inp.neumann_bc_info)
    small = [float(s.dof[2 * tip + 1]) * k / 10.0 for k in range(11)]
    inp = InpInfo(deck("beamDeflec_quadPSE_largeD_load800.inp"))
    s = oracle_system_from_inp(inp)
    large, newton = [0.0], [0]
    advance = s.advance_inc

    def recording(bcs):
        ok, loops = advance(bcs)
        if ok and s.time1 > 0.1 * len(large) - 0.05:
            large.append(float(s.dof[2 * tip + 1]))
            newton.append(int(loops))
        return ok, loops
    s.advance_inc = recording
    s.solve(dict(inp.time_incs, ini_inc=0.1, max_inc=0.1), inp.dirichlet_bc_info, inp.neumann_bc_info)
    assert len(large) == 11 and all(i["converged"] for i in s.increments)
    return tip, small, large, newton, s.n_solves


def main():
    rgba = np.asarray(Image.open(SRC).convert("RGBA")).astype(float)
    a = rgba[..., 3:4] / 255.0
    im = rgba[..., :3] * a + 255.0 * (1.0 - a)                      # on white
    darkness = 255.0 - im.mean(axis=2)
    H, W = darkness.shape
    # the spines: the long dark rows / columns
    rows = np.where((darkness > 128).sum(axis=1) > 0.6 * W)[0]
    cols = np.where((darkness > 128).sum(axis=0) > 0.6 * H)[0]
    bottom, left = rows.max(), cols.min()
    # tick marks: short dark strokes just outside the bottom / left spine
    band = darkness[bottom + 4:bottom + 8, :].mean(axis=0)
    lab, n = ndimage.label(band > 100)
    xt = [ndimage.center_of_mass(band, lab, i + 1)[0] for i in range(n)]
    band = darkness[:, left - 9:left - 4].mean(axis=1)
    lab, n = ndimage.label(band > 100)
    yt = [ndimage.center_of_mass(band, lab, i + 1)[0] for i in range(n)]
    assert len(xt) == 5 and len(yt) == 4, (xt, yt)                  # 0, 200, ..., 800 MPa; 60, 40, 20, 0
    px = np.polyfit([0, 200, 400, 600, 800], xt, 1)                 # pixel column of a load
    py = np.polyfit([60, 40, 20, 0], yt, 1)                         # pixel row of a deflection
    assert np.abs(np.polyval(px, [0, 200, 400, 600, 800]) - xt).max() < 0.8     # (ticks one or two pixels wide)
    assert np.abs(np.polyval(py, [60, 40, 20, 0]) - yt).max() < 0.6

    def is_colour(c, tol=90):
        return np.abs(im - np.array(c, dtype=float)).sum(axis=2) < tol

    masks = {k: is_colour(c) for k, c in COLOURS.items()}
    masks["abaqus"] = is_colour(ABAQUS)
    white = im.sum(axis=2) > 700
    # the markers' radius from the clean blue discs (vertical extent at their own column, anti-aliased edge included)
    radii = []
    for k in range(3, 11):
        x = int(round(np.polyval(px, 80 * k)))
        ys = np.where(masks["small_deformation"][:, x])[0]
        ys = ys[ys < np.polyval(py, 6.0 * k)]                       # (above the other two curves)
        if ys.size:                                                 # (the point at 560 MPa lies under the legend)
            radii.append((ys.max() - ys.min() + 2) / 2.0)
    R = float(np.median(radii))
    yy, xx = np.mgrid[0:H, 0:W]

    def centre(series, k, guess_rows):
        x0 = np.polyval(px, 80 * k)
        best, arg = -1e9, None
        for cy in np.arange(guess_rows[0], guess_rows[1], 0.25):
            for cx in np.arange(x0 - 1.5, x0 + 1.51, 0.25):
                y0, y1 = int(cy - R - 2), int(cy + R + 3)
                x0i, x1i = int(cx - R - 2), int(cx + R + 3)
                inside = (yy[y0:y1, x0i:x1i] - cy) ** 2 + (xx[y0:y1, x0i:x1i] - cx) ** 2 <= (R - 0.5) ** 2
                score = (masks[series][y0:y1, x0i:x1i] & inside).sum() - 2.0 * (white[y0:y1, x0i:x1i] & inside).sum()
                if score > best:
                    best, arg = score, (cx, cy)
        return arg, best

    out = {"source": "README.assets/load-deflection-curve.png (README.md:95, Fig. 2 (d)), 938 x 745 pixels",
           "load_MPa": [80.0 * k for k in range(11)],
           "pixels_per_MPa": float(px[0]), "pixels_per_unit_deflection": float(-py[0]), "marker_radius_px": R,
           "reading_uncertainty": "about +- 0.75 pixel = +- 0.1 of deflection (quarter-pixel search, anti-aliased edges, "
                                  "covered discs)"}
    for series in COLOURS:
        vals, pix, vis = [], [], []
        for k in range(11):
            # rows where this colour occurs near the marker's column
            x = int(round(np.polyval(px, 80 * k)))
            ys = np.where(masks[series][:bottom - 2, x - 3:x + 4].any(axis=1))[0]
            ys = ys[ys > 230] if k < 3 else ys                      # (the legend's samples sit above the first points)
            if ys.size == 0:                                        # fully covered by a later series
                vals.append(None)
                pix.append(None)
                vis.append(0.0)
                continue
            (cx, cy), score = centre(series, k, (ys.min() - R, ys.max() + R))
            frac = score / (np.pi * (R - 0.5) ** 2)
            if frac < 0.35:                                         # only a sliver visible: no reading
                vals.append(None)
                pix.append(None)
                vis.append(round(float(max(frac, 0.0)), 2))
                continue
            vals.append(round(float((cy - py[1]) / py[0]), 3))
            pix.append([round(float(cx), 2), round(float(cy), 2)])
            vis.append(round(float(frac), 2))
        out[series] = vals
        out[series + "_pixels"] = pix
        out[series + "_visible_fraction"] = vis
    # the Abaqus series: discs = what is left of the green mask after eroding the connecting line away
    g = masks["abaqus"].copy()
    g[:int(np.polyval(py, 31.0)), :] = False                        # (the legend)
    core = ndimage.binary_erosion(g, structure=np.ones((5, 5)))
    lab, n = ndimage.label(ndimage.binary_dilation(core, structure=np.ones((5, 5))) & g)
    pts = sorted(ndimage.center_of_mass(g, lab, i + 1)[::-1] for i in range(n))
    out["large_deformation_abaqus"] = [[round(float((cx - px[1]) / px[0]), 1), round(float((cy - py[1]) / py[0]), 3)]
                                       for cx, cy in pts]
    tip, small, large, newton, solves = oracle_curves()
    out["oracle"] = {"tip_node": tip, "small_deformation": small, "large_deformation": large,
                     "newton_loops": newton, "linear_solves": solves}
    boxes = gif_boxes()
    assert len(boxes) == 21 and boxes[0] == [316, 32]
    ob, on, osolves = oracle_boxes()
    out["stable_gif"] = {"source": "tests/beam_deflection/load800_freeEnd_largeDef/beamDeflec_quadPSE_largeD_load800_stable.gif, "
                                   "21 frames of 512 x 512", "load_MPa": [40.0 * k for k in range(21)],
                         "pixels_per_unit": boxes[0][0] / 40.0, "box_pixels": boxes,
                         "oracle_box": ob, "oracle_newton_loops": on, "oracle_linear_solves": osolves}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    for series in list(COLOURS) + ["large_deformation_abaqus"]:
        print(series, out[series], out.get(series + "_visible_fraction", ""))
    print("oracle", out["oracle"])
    sc = out["stable_gif"]["pixels_per_unit"]
    print("gif frames: worst |oracle - picture| of the box:",
          max(max(abs(o[0] - b[0] / sc), abs(o[1] - b[1] / sc)) for o, b in zip(ob, boxes)))


if __name__ == "__main__":
    main()
