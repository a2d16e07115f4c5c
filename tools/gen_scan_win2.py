"""Generates the inline-asm column blocks of the two-step ("window of two") form of the scan kernels
(the section between the GENERATED markers of pioran.jl_amd/csrc/celerite_scan.hip).  Usage: python tools/gen_scan_win2.py"""
import re
from pathlib import Path

SRC = Path(__file__).resolve().parents[1] / "pioran.jl_amd" / "csrc" / "celerite_scan.hip"


def block(rpl: int) -> str:
    R = range(rpl)
    T = ", ".join(f'[t{i}] "+v"(T[{i}])' for i in R)
    Tin = ", ".join(f'[t{i}] "v"(T[{i}])' for i in R)
    ra = ", ".join(f'[a{i}] "+v"(rA[{i}])' for i in R)
    rb = ", ".join(f'[b{i}] "+v"(rB[{i}])' for i in R)
    ha = ", ".join(f'[g{i}] "v"(hA[{i}])' for i in R)
    mb = ", ".join(f'[m{i}] "v"(mB[{i}])' for i in R)
    ph = ", ".join(f'[h{i}] "v"(ph[{i}])' for i in R)
    ppo = ", ".join(f'[p{i}] "=&v"(pp[{i}])' for i in R)
    ppi = ", ".join(f'[p{i}] "v"(pp[{i}])' for i in R)
    mv = " ".join(f"PD_FMAC(a{i}, ua, t{i})" for i in R) + " " + " ".join(f"PD_FMAC(b{i}, ub, t{i})" for i in R)
    pm = " ".join(f"PD_MUL(p{i}, h{i}, pk)" for i in R)
    sc = " ".join(f"PD_MUL(t{i}, p{i}, t{i})" for i in R)
    r1 = " ".join(f"PD_FMAC(t{i}, wa, g{i})" for i in R) + " " + " ".join(f"PD_FMAC(t{i}, wb, m{i})" for i in R)
    n = rpl
    return f'''template <int N>
struct MatVec2<{n}, N> {{   // rA += T bcast(u~A_k), rB += T bcast(u~B_k) for one column k of this lane's rows
    static __device__ __forceinline__ void run(const double (&T)[{n}], double (&rA)[{n}], double (&rB)[{n}], double ua, double ub)
    {{
        asm volatile({mv}
                     : {ra}, {rb}
                     : {Tin}, [ua] "v"(ua), [ub] "v"(ub), [n] "i"(N));
    }}
}};
template <int N>
struct Update2<{n}, N> {{   // T = (phAB_i bcast(phAB_k)) T + hA_i bcast(wA_k) + mB_i bcast(wB_k)
    static __device__ __forceinline__ void run(double (&T)[{n}], const double (&hA)[{n}], const double (&mB)[{n}], const double (&ph)[{n}],
                                               double wa, double wb, double phs)
    {{
        double pk, pp[{n}];
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\\n\\t"
                     {pm} {sc} {r1}
                     : {T}, [pk] "=&v"(pk), {ppo}
                     : {ha}, {mb}, {ph}, [wa] "v"(wa), [wb] "v"(wb), [phs] "v"(phs), [n] "i"(N));
    }}
}};
template <int N>
struct Update2First<{n}, N> {{   // first column of a (cos, sin) pair: also hands phAB_i phAB_k to the second
    static __device__ __forceinline__ void run(double (&T)[{n}], const double (&hA)[{n}], const double (&mB)[{n}], const double (&ph)[{n}],
                                               double wa, double wb, double phs, double (&pp)[{n}])
    {{
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\\n\\t"
                     {pm} {sc} {r1}
                     : {T}, [pk] "=&v"(pk), {ppo}
                     : {ha}, {mb}, {ph}, [wa] "v"(wa), [wb] "v"(wb), [phs] "v"(phs), [n] "i"(N));
    }}
}};
template <int N>
struct Update2Second<{n}, N> {{
    static __device__ __forceinline__ void run(double (&T)[{n}], const double (&hA)[{n}], const double (&mB)[{n}], double wa, double wb,
                                               const double (&pp)[{n}])
    {{
        asm volatile({sc} {r1}
                     : {T}
                     : {ha}, {mb}, {ppi}, [wa] "v"(wa), [wb] "v"(wb), [n] "i"(N));
    }}
}};
'''


def block3(rpl: int) -> str:
    """column blocks of the THREE-step form (window of three): MatVec3, Update3, Update3First, Update3Second"""
    R = range(rpl)
    T = ", ".join(f'[t{i}] "+v"(T[{i}])' for i in R)
    Tin = ", ".join(f'[t{i}] "v"(T[{i}])' for i in R)
    ra = ", ".join(f'[a{i}] "+v"(rA[{i}])' for i in R)
    rb = ", ".join(f'[b{i}] "+v"(rB[{i}])' for i in R)
    rc = ", ".join(f'[c{i}] "+v"(rC[{i}])' for i in R)
    ha = ", ".join(f'[g{i}] "v"(hA[{i}])' for i in R)
    hb = ", ".join(f'[k{i}] "v"(hB[{i}])' for i in R)
    mc = ", ".join(f'[m{i}] "v"(mC[{i}])' for i in R)
    ph = ", ".join(f'[h{i}] "v"(ph[{i}])' for i in R)
    ppo = ", ".join(f'[p{i}] "=&v"(pp[{i}])' for i in R)
    ppi = ", ".join(f'[p{i}] "v"(pp[{i}])' for i in R)
    mv = " ".join(f"PD_FMAC(a{i}, ua, t{i})" for i in R) + " " + " ".join(f"PD_FMAC(b{i}, ub, t{i})" for i in R) + " " + " ".join(f"PD_FMAC(c{i}, uc, t{i})" for i in R)
    pm = " ".join(f"PD_MUL(p{i}, h{i}, pk)" for i in R)
    sc = " ".join(f"PD_MUL(t{i}, p{i}, t{i})" for i in R)
    r1 = (" ".join(f"PD_FMAC(t{i}, wa, g{i})" for i in R) + " " + " ".join(f"PD_FMAC(t{i}, wb, k{i})" for i in R) + " "
          + " ".join(f"PD_FMAC(t{i}, wc, m{i})" for i in R))
    n = rpl
    return f'''template <int N>
struct MatVec3<{n}, N> {{   // rA += T bcast(u~A_k), rB += T bcast(u~B_k), rC += T bcast(u~C_k) for one column k of this lane's rows
    static __device__ __forceinline__ void run(const double (&T)[{n}], double (&rA)[{n}], double (&rB)[{n}], double (&rC)[{n}], double ua, double ub, double uc)
    {{
        asm volatile({mv}
                     : {ra}, {rb}, {rc}
                     : {Tin}, [ua] "v"(ua), [ub] "v"(ub), [uc] "v"(uc), [n] "i"(N));
    }}
}};
template <int N>
struct Update3<{n}, N> {{   // T = (cC_i bcast(cC_k)) T + hA_i bcast(wA_k) + hB_i bcast(wB_k) + mC_i bcast(wC_k)
    static __device__ __forceinline__ void run(double (&T)[{n}], const double (&hA)[{n}], const double (&hB)[{n}], const double (&mC)[{n}], const double (&ph)[{n}],
                                               double wa, double wb, double wc, double phs)
    {{
        double pk, pp[{n}];
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\\n\\t"
                     {pm} {sc} {r1}
                     : {T}, [pk] "=&v"(pk), {ppo}
                     : {ha}, {hb}, {mc}, {ph}, [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [phs] "v"(phs), [n] "i"(N));
    }}
}};
template <int N>
struct Update3First<{n}, N> {{   // first column of a (cos, sin) pair: also hands cC_i cC_k to the second
    static __device__ __forceinline__ void run(double (&T)[{n}], const double (&hA)[{n}], const double (&hB)[{n}], const double (&mC)[{n}], const double (&ph)[{n}],
                                               double wa, double wb, double wc, double phs, double (&pp)[{n}])
    {{
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\\n\\t"
                     {pm} {sc} {r1}
                     : {T}, [pk] "=&v"(pk), {ppo}
                     : {ha}, {hb}, {mc}, {ph}, [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [phs] "v"(phs), [n] "i"(N));
    }}
}};
template <int N>
struct Update3Second<{n}, N> {{
    static __device__ __forceinline__ void run(double (&T)[{n}], const double (&hA)[{n}], const double (&hB)[{n}], const double (&mC)[{n}], double wa, double wb, double wc,
                                               const double (&pp)[{n}])
    {{
        asm volatile({sc} {r1}
                     : {T}
                     : {ha}, {hb}, {mc}, {ppi}, [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [n] "i"(N));
    }}
}};
'''


def main():
    text = SRC.read_text()
    gen = ("// ---- GENERATED by tools/gen_scan_win2.py: column blocks of the two-step form (do not edit by hand) ----\n"
           "template <int RPL, int N> struct MatVec2;\ntemplate <int RPL, int N> struct Update2;\n"
           "template <int RPL, int N> struct Update2First;\ntemplate <int RPL, int N> struct Update2Second;\n"
           + "".join(block(r) for r in range(1, 6))
           + "template <int RPL, int N> struct MatVec3;\ntemplate <int RPL, int N> struct Update3;\n"
             "template <int RPL, int N> struct Update3First;\ntemplate <int RPL, int N> struct Update3Second;\n"
           + "".join(block3(r) for r in range(1, 4)) + "// ---- END GENERATED ----\n")
    pat = re.compile(r"// ---- GENERATED by tools/gen_scan_win2\.py.*?// ---- END GENERATED ----\n", re.S)
    if pat.search(text):
        text = pat.sub(lambda m: gen, text)
    else:
        text = text.replace("#undef PD_FMAC\n#undef PD_MUL\n", gen + "#undef PD_FMAC\n#undef PD_MUL\n")
    SRC.write_text(text)


if __name__ == "__main__":
    main()
