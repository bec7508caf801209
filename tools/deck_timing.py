"""time a whole deck through System_of_equations with PCG hipGraph on/off and different poll intervals."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from femcy_amd import backend as be
from femcy_amd.body import Body
from femcy_amd.reader import InpInfo
from femcy_amd.stiffnessMtrx import System_of_equations

deck = sys.argv[1]
for graph, poll in ((0, 32), (1, 32), (0, 128), (1, 128), (1, 512)):
    inp = InpInfo(deck)
    body = Body(nodes=inp.nodes, elements=list(inp.eSets.values())[0], ELE=inp.ELE)
    s = System_of_equations(body, list(inp.materials.values())[0], inp.geometric_nonlinear, verbose=False)
    s.ctx.set_option(be.OPT_PCG_GRAPH, graph)
    s.ctx.set_option(be.OPT_PCG_POLL, poll)
    t = time.perf_counter()
    s.solve(inp)
    s.ctx.sync()
    dt = time.perf_counter() - t
    print(f"graph={graph} poll={poll}: {dt:.2f} s, {s.stats}, {dt/max(s.stats['cg_iterations'],1)*1e6:.1f} us per CG iteration overall")
    s.ctx.close()
