"""numpy model of the int8-shadow selection bound: quantises a sample of the synthetic corpus and its
queries as shadow_i8.hip does, checks the Cauchy-Schwarz bound |q.x - s_r s_q x8.q8| <= s_r s_q [(|x8|+|ex|)|eq| + |ex||q8|]
on every (query, row) pair, and reports delta / sigma of the score distribution and how many rows fall inside the margin
of the k-th best (vs the looser l1 form 0.5|x8|_1 + 0.5|q8|_1 + d/4).  usage: python scripts/model_int8_bound.py [dim]"""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
import _synth as O  # the library's own generator (scripts/_synth.py)
n, dim, nq, k = 400_000, int(sys.argv[1]) if len(sys.argv)>1 else 768, 16, 100
rows = O.synth_rows(0x4D564631, 0, n, dim, 0).astype(np.float64)
q = O.synth_queries(0x4D564632, nq, dim, 0).astype(np.float64)
sr = np.abs(rows).max(1)/127; x8 = np.rint(rows/sr[:,None]); ex = rows/sr[:,None]-x8
X2 = np.linalg.norm(x8,axis=1); EX = np.linalg.norm(ex,axis=1)
xn = np.linalg.norm(rows,axis=1)
for metric in ("ip","cos","l2"):
    infl=[]; infl_l1=[]
    for i in range(nq):
        sq = np.abs(q[i]).max()/127; q8=np.rint(q[i]/sq); eq=q[i]/sq-q8
        EQ=np.linalg.norm(eq); Q2=np.linalg.norm(q8)
        dot = rows@q[i]; dot8 = sr*sq*(x8@q8)
        bound = sr*sq*((X2+EX)*EQ + EX*Q2)            # per-row C-S bound
        bound_l1 = sr*sq*(0.5*np.abs(x8).sum(1)+0.5*np.abs(q8).sum()+0.25*dim)
        assert (np.abs(dot-dot8) <= bound*(1+1e-9)).all()
        qn=np.linalg.norm(q[i])
        if metric=="ip": s=dot; st=dot8; d_=bound.max(); d1=bound_l1.max()
        elif metric=="cos": s=dot/(qn*xn); st=dot8/(qn*xn); d_=(bound/(qn*xn)).max(); d1=(bound_l1/(qn*xn)).max()
        else: s=-(qn*qn+xn*xn-2*dot); st=-(qn*qn+xn*xn-2*dot8); d_=2*bound.max(); d1=2*bound_l1.max()
        # scale k to this n as if corpus were 25x bigger: use k_eff = k*n/10e6 -> ~4
        for keff in (4,):
            vk = np.sort(st)[-keff]
            kept = (st >= vk - 2*d_).sum(); kept1=(st >= vk-2*d1).sum()
            infl.append(kept/keff); infl_l1.append(kept1/keff)
        sig = s.std()
    print(metric, "dim",dim,"delta/sigma C-S", d_/sig, " l1", d1/sig, " kept/k C-S median", np.median(infl), "max", np.max(infl), " l1 median", np.median(infl_l1), " actual err std/sigma", np.std(s-st)/sig)
