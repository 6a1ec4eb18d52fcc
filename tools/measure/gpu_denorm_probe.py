import numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from siammask_amd import ops
# fp16 subnormal operands through the MFMA kernels: x = 3e-6 (subnormal in fp16: 50 * 2^-24), w = 1 -> y = K * x if the matrix pipe keeps denormal inputs
for val in (3e-6, 2e-7, 3e-5):
    x = np.full((1, 64, 8, 8), val, np.float32)
    w = np.ones((64, 64, 1, 1), np.float32)
    xq = float(np.float16(val))
    for algo, tile in (("mfma", None), ("wreg", (64, 64))):
        y = ops.conv2d(torch.from_numpy(x).cuda(), w, None, dtype="f16", algo=algo, tile=tile).cpu().numpy()
        print("x = %.3g (fp16 %.6g, subnormal %s): %s y[0] = %.6g, expected %.6g" % (val, xq, xq < 6.1e-5, algo, y.reshape(-1)[0], 64 * xq))
    # subnormal WEIGHTS too
    y = ops.conv2d(torch.from_numpy(np.ones((1, 64, 8, 8), np.float32)).cuda(), np.full((64, 64, 1, 1), val, np.float32), None, dtype="f16").cpu().numpy()
    print("   weights = %.3g: y[0] = %.6g, expected %.6g" % (val, y.reshape(-1)[0], 64 * xq))
