import sys, time, numpy as np
sys.path.insert(0,'dark-archon_amd'); sys.path.insert(0,'tests')
import pyarchon as A, archon_synth as S, oracle_binding
O = oracle_binding.Oracle()
def log(*a): print(*a, flush=True)
log("devices", A.device_count())
x = S.gen_random(1<<16)
log("hist ok", (A.hist256(x)==O.hist256(x)[0]).all())
for s in (b"abracadabra", b"aaaa", b"a", b"mississippi"):
    xx=np.frombuffer(s,np.uint8); sa,b,base=A.forward(xx); log(s, list(sa), b.tobytes(), base, (sa==O.sa(xx)).all())
for shape in S.SHAPES:
    for n in (1000, 65536, 1<<20):
        x = S.gen_shape(shape,n); t=time.time(); sa,b,base=A.forward(x); dt=time.time()-t
        P,B,b0=O.forward(x); st=A.stats()
        log(shape,n,"sa_ok",(sa==P).all(),"bwt_ok",(b==B).all() and base==b0, "%.1fms"%(dt*1e3), {k:(round(v,3) if isinstance(v,float) else v) for k,v in st.items() if v})
        out=A.inverse(B,b0); log("   inverse ok",(out==x).all(), {k:(round(v,3) if isinstance(v,float) else v) for k,v in A.stats().items() if v})
        log("   validate", A.validate(x,P))
