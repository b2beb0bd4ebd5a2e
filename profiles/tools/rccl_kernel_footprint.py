"""What do RCCL's send / recv kernels need beside the brick launch?  A communicator of ONE rank (all a one-GPU box
allows), hq_comm_selftest = the grouped ncclRecv + ncclSend the halo exchange issues; run under
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 profiles/tools/rccl_kernel_footprint.py
and read VGPR / LDS / workgroup size of the ncclDevKernel rows from the kernel trace."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hercules_amd as ha
from oracle import herc_oracle as ho

elem_ijk, lnid, node_ijk = ho.uniform_mesh(4, 4, 4)
edata = np.empty((len(lnid), 4), np.float32)
edata[:] = (62.5, 6000.0, 3464.0, 2700.0)
et, nt = ho.solver_init(lnid, edata, ho.face_bits(elem_ijk, 4, 4, 4), len(node_ijk), 1e-3, 5.0)
s = ha.Solver(lnid, et, nt, 1e-3, node_xyz=(np.asarray(node_ijk, np.int64) << 20).astype(np.int32))
s.comm_init(ha.capi.comm_unique_id())
for _ in range(3):
    s.comm_selftest(3 * 65536)
s.close()
