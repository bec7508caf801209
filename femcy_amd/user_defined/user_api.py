"""user-defined Dirichlet boundary condition hook (drop-in for
/root/reference/user_defined/user_api.py:6-30): the node set is rotated about the z axis through
(40, 5, 0) by time*pi and the `dm_specified` component of the resulting displacement is
prescribed.  The hook computes the (tiny) value list on the host and writes it into the
device-resident dof vector; users edit this file exactly as they would edit the reference's."""
import math

import numpy as np


def user_dirichletBC_values(dof_host, nodeSet, dm, dm_specified, nodes, time):
    """numpy body: writes the prescribed entries into `dof_host` and returns (indices, values)."""
    pi = 3.141592653589793
    center = np.array([40., 5., 0.])
    angle = time * pi
    rota = np.array([[math.cos(angle), math.sin(angle), 0.],
                     [-math.sin(angle), math.cos(angle), 0.],
                     [0., 0., 1.]])
    ns = np.asarray(nodeSet, dtype=np.int64)
    X = np.asarray(nodes)[ns]
    disp = (X - center) @ rota.T + center - X
    idx = ns * dm + dm_specified
    vals = disp[:, dm_specified]
    if dof_host is not None:
        dof_host[idx] = vals
    return idx, vals


def user_dirichletBC(dof, nodeSet, dm, dm_specified, nodes, time):
    """same signature as the reference kernel; `dof` is a femcy_amd DeviceVector (or a numpy array)."""
    node_ids = nodeSet.to_numpy() if hasattr(nodeSet, "to_numpy") else np.asarray(nodeSet)
    coords = nodes.to_numpy() if hasattr(nodes, "to_numpy") else np.asarray(nodes)
    if hasattr(dof, "ctx"):
        idx, vals = user_dirichletBC_values(None, node_ids, dm, dm_specified, coords, time)
        dof.ctx.scatter(dof.id, idx, vals)
    else:
        user_dirichletBC_values(dof, node_ids, dm, dm_specified, coords, time)
