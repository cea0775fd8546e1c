"""Measurement aid (not part of the product): what the vendor GEMM library reaches on plain GEMMs of
the same M x N x K as the GEMM-bound ResNet-50 convolutions (im2col-free upper bound: no gather,
no padding).  Gives the practical ceiling the igemm kernel is compared with in DESIGN.md."""
import torch

SHAPES = [  # (M, N, K, label)
    (50176, 256, 2304, '256,14->256 3x3'), (12544, 512, 4608, '512,7->512 3x3'),
    (50176, 256, 1024, '1024,14->256 1x1'), (50176, 1024, 256, '256,14->1024 1x1'),
    (200704, 128, 1152, '128,28->128 3x3'), (802816, 64, 576, '64,56->64 3x3'),
    (12544, 2048, 512, '512,7->2048 1x1'), (12544, 512, 2048, '2048,7->512 1x1'),
    (8192, 8192, 8192, 'square 8k'),
]


def main():
    dev = torch.device('cuda', 0)
    for M, N, K, label in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            c = a @ b.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 20
        e0.record()
        for _ in range(it):
            c = a @ b.t()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        print('%-22s M=%7d N=%5d K=%5d  %.3f ms  %7.1f TF/s' % (label, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))


if __name__ == '__main__':
    main()
