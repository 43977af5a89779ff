"""Parameter manifests of the block-level entry points (shared by tools/race_screen.py and tools/mx_check.py)."""


def res_manifest(cin, cout):
    m = [("in_layers.0.weight", (cin,)), ("in_layers.0.bias", (cin,)), ("in_layers.2.weight", (cout, cin, 3, 3)), ("in_layers.2.bias", (cout,)),
         ("emb_layers.1.weight", (cout, 512)), ("emb_layers.1.bias", (cout,)), ("out_layers.0.weight", (cout,)), ("out_layers.0.bias", (cout,)),
         ("out_layers.3.weight", (cout, cout, 3, 3)), ("out_layers.3.bias", (cout,))]
    if cin != cout:
        m += [("skip_connection.weight", (cout, cin, 1, 1)), ("skip_connection.bias", (cout,))]
    return m


def attn_manifest(ch):
    return [("norm.weight", (ch,)), ("norm.bias", (ch,)), ("qkv.weight", (3 * ch, ch, 1)), ("qkv.bias", (3 * ch,)), ("proj_out.weight", (ch, ch, 1)),
            ("proj_out.bias", (ch,))]
