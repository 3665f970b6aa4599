"""Is the library's fused GEMM+bias+GELU epilogue (torch._addmm_activation -> hipBLASLt) the erf GELU the reference uses or
the tanh approximation?  Compares its bf16 output bit-for-bit with bf16(gelu_erf(u)) and bf16(gelu_tanh(u)), u = the fp32
GEMM result."""
import torch
torch.manual_seed(0)
M, K, N = 40000, 768, 3072
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.04).to(torch.bfloat16)
b = torch.randn(N, device="cuda").to(torch.bfloat16)
u = torch.addmm(b.float(), x.float(), w.float().t())
y = torch._addmm_activation(b, x, w.t(), use_gelu=True)
ge = torch.nn.functional.gelu(u).to(torch.bfloat16)
gt = torch.nn.functional.gelu(u, approximate="tanh").to(torch.bfloat16)
n = y.numel()
print("bit-equal to bf16(erf gelu): %.4f %%   to bf16(tanh gelu): %.4f %%   erf vs tanh themselves: %.4f %%" % (
    100.0 * (y == ge).sum().item() / n, 100.0 * (y == gt).sum().item() / n, 100.0 * (ge == gt).sum().item() / n))
sel = ge != gt
print("where the two differ (%d elements): fused == erf %.2f %%, fused == tanh %.2f %%" % (
    sel.sum().item(), 100.0 * (y[sel] == ge[sel]).float().mean().item(), 100.0 * (y[sel] == gt[sel]).float().mean().item()))
