"""The parameter table of the reference-side stub in INTEGRATION.md section 2, from the selector itself.

    python tools/make_integration_stub_table.py            # prints the table
    python tools/make_integration_stub_table.py --write    # rewrites it in INTEGRATION.md, between the stub-table markers

Every row is what `params.choose_params(p_max, norm2_max, glwe_dims=DEFAULT_GLWE_DIMS)` returns (128-bit noise floors, 6 sigma; GLWE
dimension 3 at N = 512 or 2 at N = 1024 where they reach the margin -- the executor's own default): it holds
6 sigma at its own bounds, hence for every program with fbs_size <= p_max and norm2_linprod <= norm2_max (a smaller p widens
the box, a smaller norm shrinks the noise).  tests/test_integration_stub.py holds INTEGRATION.md to this output and checks
the margins, so the stub cannot go stale against the noise model again."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (p_max, norm2_max): the reference's own comparison point p = 4 (experiments/analyse_results.py:317); p = 7 / 8; p = 15 at the
# adder's and multiplier's norms and at trivium's 281; p = 17 (analyse_results.py:353); p = 31 (BASELINE configs[4])
BOUNDS = [(4, 8), (8, 96), (15, 96), (17, 288), (31, 336)]
BEGIN, END = "# <stub-table>", "# </stub-table>"


def rows():
    from tfhe_fbs_map_amd.params import DEFAULT_GLWE_DIMS, choose_params
    out = {}
    for p, norm2 in BOUNDS:
        s = choose_params(p, norm2, min_margin=6.0, security=128, glwe_dims=DEFAULT_GLWE_DIMS)   # what ExecConfig() asks for
        out[(p, norm2)] = (s.n, s.log_n_poly, s.k, s.l_bsk, s.beta_bsk, s.t_ksk, s.gamma_ksk, s.bsk_group)
    return out


def table_text(indent=" " * 8):
    body = ",\n".join("%s        %r: %r" % (indent, k, v) for k, v in rows().items())
    return ("%s%s  written by tools/make_integration_stub_table.py from tfhe_fbs_map_amd.params.choose_params: 128-bit noise,\n"
            "%s# 6 sigma at the row's own bounds.  (p_max, norm2_max): (n, log2 N, k, l, beta, t, gamma, key bits per blind-rotation step)\n"
            "%sSETS = {\n%s}\n%s%s\n" % (indent, BEGIN, indent, indent, body, indent, END))


def main():
    text = table_text()
    if "--write" not in sys.argv:
        print(text, end="")
        return
    path = os.path.join(ROOT, "INTEGRATION.md")
    doc = open(path).read()
    new = re.sub(r"[ ]*%s.*?%s\n" % (re.escape(BEGIN), re.escape(END)), lambda _: text, doc, flags=re.S)
    assert new != doc or text in doc, "markers not found in INTEGRATION.md"
    open(path, "w").write(new)
    print("rewrote the stub table in", path)


if __name__ == "__main__":
    main()
