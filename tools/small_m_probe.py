import os,sys,time
sys.path[:0]=[".","naqs-for-quantum-chemistry_amd"]
import torch
sys.argv=["x"]
import bench
from naqs_amd import hamiltonian, packing
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
from naqs_amd.fused import FusedLogPsi
dev=torch.device("cuda",0)
ham_p=packing.load_packed("tests/golden/ham_N2.npz")
hil=Hilbert.get(20,7,7,encoding=Encoding.SIGNED)
wf=NAQSComplex_NADE_orbitals(hil,device=dev,qubit_ordering=-1,amp_hidden_size=[64],phase_hidden_size=[512,512],use_amp_spin_sym=True,use_phase_spin_sym=False,aggregate_phase=False,n_alpha_electrons=7,n_beta_electrons=7)
f=FusedLogPsi(wf)
out=[]
for M in (300,1255,4000):
    k,_,_=bench.make_batch(ham_p,M,0); keys=hamiltonian.keys_to_device(k,dev)
    for _ in range(20): f.log_psi(keys)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(500): f.log_psi(keys)
    torch.cuda.synchronize(); a=(time.perf_counter()-t)/500*1e6
    for _ in range(20): f.forward_saved(keys)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(500): f.forward_saved(keys)
    torch.cuda.synchronize(); b=(time.perf_counter()-t)/500*1e6
    out.append(f"M {M}: log_psi {a:.1f} us, train forward {b:.1f} us")
print(" | ".join(out))
