#!/usr/bin/env python3
"""Phase timeline of the EXACT temporal attention kernel from a -DD3DP_ATTN_STAMP=1 build (tools/build_variant.sh stamp
attention.hip "-DD3DP_ATTN_STAMP=1"; run with D3DP_LIB=.../libd3dp_stamp.so): cycles between the kernel's barriers."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from d3dp_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    raw = ctypes.CDLL(os.environ["D3DP_LIB"])
    F, J, C, heads, seqs = 243, 17, 512, 8, 31
    T = seqs * F * J
    qkv = torch.randn(T, 3 * C, device="cuda")
    out = torch.empty(T, C, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(lib.d3dp_op_attention(0, 2, 1, qkv.data_ptr(), out.data_ptr(), seqs, F, J, C, heads, st))
    torch.cuda.synchronize()
    buf = np.zeros(4 * 8 * 16 * 8, dtype=np.uint64)
    assert raw.d3dp_debug_attn_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    s = buf.reshape(4, 8, 16, 8).astype(np.int64)
    names = ["wait K,q (vmcnt)", "barrier 1", "scores + softmax", "wait V (vmcnt 0)", "barrier 2", "issue K', P.V, store O",
             "barrier 3", "issue V' -> next top"]
    for wg in range(2):
        for wave in (0, 4, 7):
            t = s[wg, wave]
            d = np.empty((12, 8), dtype=np.int64)
            for it in range(2, 14):
                for k in range(7):
                    d[it - 2, k] = t[it, k + 1] - t[it, k]
                d[it - 2, 7] = t[it + 1, 0] - t[it, 7]
            print(f"wg {wg} wave {wave}: cycles per problem {d.sum(1).mean():8.0f}")
            for k in range(8):
                print(f"    {names[k]:28s} mean {d[:, k].mean():8.0f}  min {d[:, k].min():7d}  max {d[:, k].max():7d}")


if __name__ == "__main__":
    main()
