import sys, os; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests/golden')
import numpy as np, make_ref_golden as G
from teb_local_planner_amd import planner
g=np.load('/root/repo/tests/golden/ref_f3_hcp_ticks.npz')
for cname, case in G.hcp_tick_cases().items():
    hcp=planner.HomotopyClassPlanner(case["cfg"], case["obst"], [], None, max_tebs=8, max_poses=256)
    for t,(st,gl) in enumerate(zip(case["starts"], case["goals"])):
        sv=None if case["start_vels"] is None else case["start_vels"][t]
        hcp.plan(st, gl, sv)
        bands=hcp.bands()
        ref={k[len(cname)+2+len(str(t))+2:]: g[k] for k in g.files if k.startswith("%s__%d__"%(cname,t))}
        res=hcp.results()
        print(cname, t, 'count', len(bands), len(ref['n']), 'best', hcp.best_teb_, int(ref['best']), 'n', [len(b[0]) for b in bands], list(ref['n']))
        print('   cost', np.array(res.cost[:len(bands)]), ref['costs'])
        for k in range(min(len(bands), len(ref['n']))):
            w=G.unpack(ref,k)
            if len(w[0])==len(bands[k][0]):
                print('   band',k, max(np.abs(a-b).max() for a,b in zip(bands[k],w)))
