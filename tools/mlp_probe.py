"""Diagnostic (GPU box): what the bench's 24-256-256-2 policy MLP costs per partition tick in a few formulations."""
import torch, time
dev = torch.device('cuda:0')
for M in (5461, 16384):
    x = torch.randn(M, 24, device=dev); w1 = torch.randn(24, 256, device=dev); b1 = torch.zeros(256, device=dev)
    w2 = torch.randn(256, 256, device=dev) / 16; b2 = torch.zeros(256, device=dev); w3 = torch.randn(256, 2, device=dev) / 16; b3 = torch.zeros(2, device=dev)
    a = torch.empty(M, 2, device=dev)
    def f32():
        h1 = torch.relu(x @ w1 + b1); h2 = torch.relu(h1 @ w2 + b2); torch.tanh(h2 @ w3 + b3, out=a)
    w1t, w2t, w3t = w1.t().contiguous(), w2.t().contiguous(), w3.t().contiguous()
    def lin():
        h1 = torch.relu(torch.nn.functional.linear(x, w1t, b1)); h2 = torch.relu(torch.nn.functional.linear(h1, w2t, b2)); torch.tanh(torch.nn.functional.linear(h2, w3t, b3), out=a)
    def nobias():
        h1 = torch.relu((x @ w1).add_(b1)); h2 = torch.relu((h1 @ w2).add_(b2)); torch.tanh((h2 @ w3).add_(b3), out=a)
    xb, w1b, w2b, w3b = x.bfloat16(), w1.bfloat16(), w2.bfloat16(), w3.bfloat16()
    def bf16():
        h1 = torch.relu((x.bfloat16() @ w1b).add_(b1)); h2 = torch.relu((h1 @ w2b).add_(b2)); torch.tanh(((h2 @ w3b).float()).add_(b3), out=a)
    for nm, f in (('x @ w + b (bench)', f32), ('F.linear', lin), ('matmul then add_', nobias), ('bf16 matmuls', bf16)):
        for _ in range(5): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): f()
        torch.cuda.synchronize(); print('M=%d %-20s %.1f us' % (M, nm, (time.perf_counter() - t) / 50 * 1e6))
