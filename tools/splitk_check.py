"""Split-K check: a deep-level ResBlock (internal NHWC convs take the split-K path) against the same block with DRM_NO_SPLITK=1."""
import os, subprocess, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drmnet_amd import ops, synth

def man(cin, cout):
    m = [("in_layers.0.weight", (cin,)), ("in_layers.0.bias", (cin,)), ("in_layers.2.weight", (cout, cin, 3, 3)), ("in_layers.2.bias", (cout,)),
         ("emb_layers.1.weight", (cout, 512)), ("emb_layers.1.bias", (cout,)), ("out_layers.0.weight", (cout,)), ("out_layers.0.bias", (cout,)),
         ("out_layers.3.weight", (cout, cout, 3, 3)), ("out_layers.3.bias", (cout,))]
    if cin != cout: m += [("skip_connection.weight", (cout, cin, 1, 1)), ("skip_connection.bias", (cout,))]
    return m

if len(sys.argv) > 1:
    ops.set_precision("f16x3"); dev = torch.device("cuda:0")
    outs = {}
    for (n, cin, cout, h, w) in [(32, 768, 768, 4, 8), (1, 768, 768, 4, 4), (32, 1536, 768, 4, 8), (3, 640, 640, 8, 8)]:
        P = [p.to(dev) for p in synth.synth_state_dict(man(cin, cout), 1).values()]
        g = torch.Generator().manual_seed(0)
        x = torch.randn((n, cin, h, w), generator=g).to(dev); emb = torch.randn((n, 512), generator=g).to(dev)
        outs[(n, cin, cout, h, w)] = ops.resblock(P, x, emb).cpu()
    torch.save(outs, sys.argv[1])
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "/tmp/sk_a.pt"], env=env)
    env["DRM_NO_SPLITK"] = "1"
    subprocess.check_call([sys.executable, __file__, "/tmp/sk_b.pt"], env=env)
    a, b = torch.load("/tmp/sk_a.pt"), torch.load("/tmp/sk_b.pt")
    for k in a:
        print(k, "split-K vs plain rel", float((a[k].double() - b[k].double()).norm() / b[k].double().norm()), "identical" if torch.equal(a[k], b[k]) else "")
