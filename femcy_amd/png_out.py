"""headless PNG of a result: the deformed mesh coloured by nodal von Mises stress (or |u|) with a colour bar --
what the reference's GGUI windows show (`Body.show` / `show2d`, body.py:100-162, colour bar :26-61) and its
README lists as future work ("write results to files").  Rendering is matplotlib's Agg canvas (no display);
the triangles are the element plugin's drawing triangles (`ELE._tri_split`, the reference's `getMesh`)."""
import numpy as np


def nodal_average(el: np.ndarray, patch_vals: np.ndarray, nn: int) -> np.ndarray:
    """element-patch nodal values [ne, npe] (ELE.extrapolate) -> one value per node (mean over the patches)."""
    s = np.bincount(el.ravel(), weights=np.asarray(patch_vals).ravel(), minlength=nn)
    c = np.bincount(el.ravel(), minlength=nn)
    return s / np.maximum(c, 1)


def render_png(path: str, nodes: np.ndarray, tris: np.ndarray, values: np.ndarray, label: str = "",
               title: str = "", edges: bool = True, dpi: int = 150):
    """nodes [nn, 2|3] (already deformed), tris [nt, 3] node ids, values [nn]."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    nodes, tris, values = np.asarray(nodes, float), np.asarray(tris, np.int64), np.asarray(values, float)
    vmin, vmax = float(values.min()), float(values.max())
    if vmax <= vmin:
        vmax = vmin + 1.0
    fig = plt.figure(figsize=(8, 6), dpi=dpi)
    if nodes.shape[1] == 2:
        ax = fig.add_subplot(111)
        tpc = ax.tripcolor(nodes[:, 0], nodes[:, 1], tris, values, shading="gouraud", cmap="jet", vmin=vmin, vmax=vmax)
        if edges and tris.shape[0] <= 20000:
            ax.triplot(nodes[:, 0], nodes[:, 1], tris, color="k", linewidth=0.2)
        ax.set_aspect("equal")
        fig.colorbar(tpc, ax=ax, label=label)
    else:
        from matplotlib import cm, colors
        from mpl_toolkits.mplot3d.art3d import Poly3DCollection
        ax = fig.add_subplot(111, projection="3d")
        norm = colors.Normalize(vmin, vmax)
        face = cm.jet(norm(values[tris].mean(axis=1)))
        ax.add_collection3d(Poly3DCollection(nodes[tris], facecolors=face, edgecolors=(0, 0, 0, 0.3) if edges else None,
                                             linewidths=0.1))
        lo, hi = nodes.min(axis=0), nodes.max(axis=0)
        ax.set_xlim(lo[0], hi[0]); ax.set_ylim(lo[1], hi[1]); ax.set_zlim(lo[2], hi[2])
        ax.set_box_aspect(np.maximum(hi - lo, 1e-12))
        fig.colorbar(cm.ScalarMappable(norm=norm, cmap="jet"), ax=ax, label=label, shrink=0.7)
    ax.set_title(title)
    fig.savefig(path)
    plt.close(fig)


def write_png(path: str, system, field: str = "mises", scale: float = 1.0):
    """`system`: a solved System_of_equations.  field = "mises" (nodal average of the extrapolated Gauss-point
    von Mises stress) or "disp" (|u|); the mesh is drawn at nodes + scale * u."""
    body, ELE, dm = system.body, system.ELE, system.dm
    el = np.asarray(body.np_elements)
    u = system.dof.to_numpy().reshape(-1, dm)
    xy = np.asarray(body.np_nodes) + scale * u
    if field == "disp":
        vals, label = np.linalg.norm(u, axis=1), "|u|"
    else:
        system.compute_strain_stress()
        system.ELE.extrapolate(system.mises_stress, system.nodal_vals)
        vals, label = nodal_average(el, np.asarray(system.nodal_vals.to_numpy()), xy.shape[0]), "von Mises stress"
    if dm == 2:
        tris = np.concatenate([el[:, list(t)] for t in ELE._tri_split], axis=0)
    else:
        _, _, tris = ELE.getMesh(el)                    # outer surface only
    render_png(path, xy, tris, vals, label=label, title="%d elements, max %s = %.6g" % (el.shape[0], label, vals.max()))
