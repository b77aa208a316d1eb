import ctypes, os, sys
ROOT="/root/repo"
os.environ["ARCHON_HIP_LIB"]=os.path.join(os.environ.get("GRAFT_REPO_ROOT",ROOT),"dark-archon_amd","libarchon_hip_exp.so")
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT",ROOT),"dark-archon_amd"))
import numpy as np, torch, pyarchon, archon_synth as S
n=64<<20
L=pyarchon.lib()
L.archon_hip_post_bound.restype = ctypes.c_size_t; L.archon_hip_post_bound.argtypes = [ctypes.c_uint32]
L.archon_hip_post_encode_dev.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
x=S.gen_shape(sys.argv[1] if len(sys.argv)>1 else "text", n); x_t=torch.from_numpy(x).cuda()
bwt=torch.empty(n,dtype=torch.uint8,device="cuda"); base=torch.zeros(1,dtype=torch.int32,device="cuda")
pyarchon.forward_dev(x_t, torch.empty(n,dtype=torch.int32,device="cuda"), bwt, base)
cap=L.archon_hip_post_bound(n); d_out=torch.empty(cap,dtype=torch.uint8,device="cuda"); got=ctypes.c_size_t(0)
torch.cuda.synchronize()
for _ in range(2): assert L.archon_hip_post_encode_dev(bwt.data_ptr(), n, d_out.data_ptr(), cap, ctypes.byref(got), 0, None)==0
buf=(ctypes.c_ulonglong*16)(); assert L.archon_hip_exp_post_stamps(buf)==0
names=["load+last","init list","mtf","mtf barrier wait","lastnz scan+hist walk","barrier","huffman","canonical","barrier", "bits+emit","barrier","copy out"]
for i in range(1,12): print("%-24s %9d" % (names[i-1], buf[i]-buf[i-1]))
print("total", buf[11]-buf[0])
