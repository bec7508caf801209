"""entry point, compatible with the reference's main.py (/root/reference/main.py:10-82):
prompt for an Abaqus .inp file, solve, print displacements.  The Taichi GGUI windows of the
reference are not available headless; results are printed (and optionally saved with --save).

    python -m femcy_amd.main                 # interactive prompt, as the reference
    python -m femcy_amd.main path/to/deck.inp [--save result.npz] [--device 0] [--quiet]
"""
import argparse
import os
import time

import numpy as np

from .body import Body
from .reader.inp_info import InpInfo
from .stiffnessMtrx import System_of_equations
from .tiGadgets import field_abs_max


def run(fileName: str, device: int = 0, verbose: bool = True, tangent: str = "reference",
        allow_2d_hyperelastic: bool = False):
    from . import distributed
    inp = InpInfo(fileName, allow_2d_hyperelastic=allow_2d_hyperelastic)
    nodes, eSets = inp.nodes, inp.eSets
    material = list(inp.materials.values())[0]
    if distributed.wanted():       # launched by torch.distributed.run: one rank per GPU, one element partition each
        return run_partitioned(inp, material, verbose, tangent)
    body = Body(nodes=nodes, elements=list(eSets.values())[0], ELE=inp.ELE)
    system = System_of_equations(body, material, inp.geometric_nonlinear, device=device, verbose=verbose,
                                 tangent=tangent)
    time0 = time.time()
    system.solve(inp, show_newton_steps=True, save2path=None)
    system.ctx.sync()
    time1 = time.time()
    print(f"\033[40;33;1m system.dof = \n{system.dof.to_numpy()}, "
          f"time for finite element computing is {time1 - time0} s \033[0m")
    system.get_elasEng()
    print(f"total elastic energy is {system.elsEng}")
    system.compute_strain_stress()
    stress = system.mises_stress.to_numpy()
    print(f"\033[35;1m max mises_stress at integration point is {stress.max()} MPa \033[0m", end="; ")
    print(f"\033[40;33;1m max dof (disp) = {field_abs_max(system.dof)} \033[0m")
    system.ELE.extrapolate(system.mises_stress, system.nodal_vals)
    print(f"\033[35;1m max nodal mises_stress = {np.asarray(system.nodal_vals).max()} \033[0m")
    print(f" solver statistics: {system.stats}")
    return inp, system


def run_partitioned(inp, material, verbose: bool = True, tangent: str = "reference"):
    """the same solve with the mesh split by element over the ranks of the job (femcy_amd/distributed.py).
    Rank 0 prints; `system.dof_global` holds the gathered displacements there (None on other ranks)."""
    from . import distributed
    system, local_deck, part = distributed.partitioned_system(inp, material, verbose, tangent)
    say = print if part.rank == 0 else (lambda *a, **k: None)
    time0 = time.time()
    system.solve(local_deck, show_newton_steps=True, save2path=None)
    system.ctx.sync()
    time1 = time.time()
    system.dof_global = distributed.gather_dof(part, system.dof.to_numpy(), inp.nodes.size)
    say(f"\033[40;33;1m {part.nranks} ranks, {system.ctx.n_global} DOF: time for finite element computing is "
        f"{time1 - time0} s \033[0m")
    system.get_elasEng()                                  # collective: every rank makes the call
    say(f"total elastic energy is {system.elsEng}")
    system.compute_strain_stress()
    umax = field_abs_max(system.dof)
    say(f"\033[40;33;1m max dof (disp) = {umax} \033[0m")
    say(f" solver statistics: {system.stats}")
    return inp, system


def main(argv=None):
    os.system("")
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("inp", nargs="?", default=None)
    ap.add_argument("--save", default=None, help="write results to this .npz, a legacy .vtk (mesh + displacement + Mises) for ParaView, or a .png picture of the deformed mesh coloured by von Mises stress")
    ap.add_argument("--device", type=int, default=int(os.environ.get("FEMCY_DEVICE", "0")))
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--allow-2d-neo-hookean", action="store_true",
                    help="accept *Hyperelastic, neo hooke on CPE elements (plane-strain extension; the reference "
                         "rejects it)")
    ap.add_argument("--tangent", choices=("reference", "consistent"), default="reference",
                    help="matrix of the Newton iterations: the reference's B^T C B with the constant C (default; parity "
                         "with FEMcy), or the consistent tangent (extension: far fewer linear solves, different iterates)")
    args = ap.parse_args(argv)
    fileName = args.inp or input("\033[32;1m please give the .inp format's input file path and name: \033[0m")
    inp, system = run(fileName, device=args.device, verbose=not args.quiet, tangent=args.tangent,
                      allow_2d_hyperelastic=args.allow_2d_neo_hookean)
    if getattr(system, "part", None) is not None:
        if args.save and system.part.rank == 0:
            np.savez(args.save, nodes=inp.nodes, dof=system.dof_global)
        return
    if args.save:
        if args.save.endswith(".vtk"):
            from .vtk_out import write_vtk
            write_vtk(args.save, system)
        elif args.save.endswith(".png"):
            from .png_out import write_png
            write_png(args.save, system)
        else:
            np.savez(args.save, nodes=inp.nodes, dof=system.dof.to_numpy(),
                     cauchy_stress=system.cauchy_stress.to_numpy(), mises_stress=system.mises_stress.to_numpy())
        print(f" saved {args.save}")


if __name__ == "__main__":
    main()
