"""Which calls make the HIP runtime print 'Cannot get amd_mem_obj for ptr' (hip_memory.cpp, error level) on a pageable host
pointer?  One subprocess per action under AMD_LOG_LEVEL=1, counting the lines on its stderr."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRE = ("import sys, importlib, numpy as np, torch\nsys.path.insert(0, %r)\n"
       "pkg = importlib.import_module('mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd')\n"
       "rng = np.random.default_rng(0); Q, K, V = rng.uniform(-1, 1, (2048, 128)), rng.uniform(-1, 1, (4096, 128)), rng.uniform(-1, 1, (4096, 128))\n"
       "torch.cuda.init(); pkg.init(1)\n" % ROOT)
ACTIONS = {
    "nothing (import, init)": "",
    "pkg.attention(Q, K, V) x 3, pageable numpy arrays": "for _ in range(3): pkg.attention(Q, K, V)\n",
    "the same with SDPA_HOST_PROBE=1": "import os\nos.environ['SDPA_HOST_PROBE']='1'\nfor _ in range(3): pkg.attention(Q, K, V)\n",
    "torch.from_numpy(K).cuda() x 3": "for _ in range(3): torch.from_numpy(K).cuda()\ntorch.cuda.synchronize()\n",
    "torch.from_numpy(K).is_pinned() x 3": "for _ in range(3): torch.from_numpy(K).is_pinned()\n",
    "HipBackend.to_device(K) x 3": "be = pkg.HipBackend('cuda:0')\nfor _ in range(3): be.to_device(K)\ntorch.cuda.synchronize()\n",
    "torch .cpu() of a device tensor x 3": "t = torch.zeros(1000, device='cuda')\nfor _ in range(3): t.cpu()\n",
}
for name, body in ACTIONS.items():
    env = dict(os.environ, AMD_LOG_LEVEL="1")
    r = subprocess.run([sys.executable, "-c", PRE + body], capture_output=True, text=True, env=env, timeout=300)
    print("%-60s rc %d  amd_mem_obj lines: %d" % (name, r.returncode, r.stderr.count("amd_mem_obj")), flush=True)
