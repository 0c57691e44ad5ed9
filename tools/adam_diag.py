"""Which elementary fp32 roundings does torch.optim.Adam (default multi-tensor path) make on this GPU?  Prints, for one step from
zero moments and for a second step, how many elements of exp_avg / exp_avg_sq / param differ from candidate evaluations
(torch elementwise ops, and float64-exact emulations rounded once).  Used to pin csrc/optim_kernels.hip."""
import torch

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(5)
n = 1 << 20
p0 = torch.randn(n, device=dev, generator=g) * torch.logspace(-6, 2, n, device=dev)
g1 = torch.randn(n, device=dev, generator=g) * torch.logspace(-8, 1, n, device=dev).flip(0)
g2 = torch.randn(n, device=dev, generator=g)
ref = torch.nn.Parameter(p0.clone())
opt = torch.optim.Adam([ref], lr=1e-3)
b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-3


def ne(a, b):
    return int((a != b).sum())


def f32(x64):
    return x64.to(torch.float32)


for t, gr in ((1, g1), (2, g2)):
    st = opt.state.get(ref, None)
    m_prev = st["exp_avg"].clone() if st else torch.zeros_like(p0)
    v_prev = st["exp_avg_sq"].clone() if st else torch.zeros_like(p0)
    p_prev = ref.data.clone()
    ref.grad = gr.clone()
    opt.step()
    st = opt.state[ref]
    m_t, v_t, p_t = st["exp_avg"], st["exp_avg_sq"], ref.data
    w1, w2 = torch.tensor(1 - b1, dtype=torch.float32).item(), torch.tensor(1 - b2, dtype=torch.float32).item()
    b2f = torch.tensor(b2, dtype=torch.float32).item()
    bc1, bc2 = 1 - b1 ** float(t), 1 - b2 ** float(t)
    neg = torch.tensor((lr / bc1) * -1, dtype=torch.float32).item()
    bc2s = torch.tensor(bc2 ** 0.5, dtype=torch.float32).item()
    epsf = torch.tensor(eps, dtype=torch.float32).item()
    D = torch.float64
    diff = gr - m_prev
    print(f"--- step {t}")
    print("m: torch.lerp", ne(torch.lerp(m_prev, gr, w1), m_t), "| m + w*diff (two roundings)", ne(m_prev + w1 * diff, m_t),
          "| fma(w, diff, m)", ne(f32(m_prev.to(D) + w1 * diff.to(D)), m_t))
    vb = v_prev * b2f
    gg = gr * gr
    print("v: vb + w2*gg (two roundings)", ne(vb + w2 * gg, v_t), "| fma(w2, gg, vb)", ne(f32(vb.to(D) + w2 * gg.to(D)), v_t),
          "| vb + (w2*g)*g", ne(vb + (w2 * gr) * gr, v_t), "| fma(w2*g, g, vb)", ne(f32(vb.to(D) + (w2 * gr).to(D) * gr.to(D)), v_t),
          "| torch.addcmul", ne(torch.addcmul(vb, gr, gr, value=1 - b2), v_t))
    sq = v_t.sqrt()
    sq_exact = f32(v_t.to(D).sqrt())
    print("sqrt: torch.sqrt vs correctly rounded", ne(sq, sq_exact))
    for name, s in (("torch.sqrt", sq), ("exact sqrt", sq_exact)):
        d_t = s / bc2s
        d_e = f32(s.to(D) / bc2s)
        d_r = s * torch.tensor(1.0 / bc2s, dtype=torch.float32).item()
        print(f"  [{name}] div: torch '/' vs correctly rounded", ne(d_t, d_e), "| vs mul by reciprocal", ne(d_t, d_r))
        for dn, d in (("torch /", d_t), ("exact /", d_e), ("x rcp", d_r)):
            de = d + epsf
            q_t = m_t / de
            q_e = f32(m_t.to(D) / de.to(D))
            for qn, q in (("torch /", q_t), ("exact /", q_e)):
                pa = p_prev + neg * q
                pf = f32(p_prev.to(D) + neg * q.to(D))
                print(f"    denom {dn:8s} quot {qn:8s}: p two roundings {ne(pa, p_t):8d} | fma {ne(pf, p_t):8d}")
    print("p: torch.addcdiv(p, m, denom, value)", ne(torch.addcdiv(p_prev, m_t, (v_t.sqrt() / bc2s) + epsf, value=(lr / bc1) * -1), p_t))
