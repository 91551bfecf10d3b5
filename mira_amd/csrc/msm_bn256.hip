// bn256: instantiates the MSM pipeline for this curve (coordinates Fq29 / FqP, scalars FrP).
#include "msm_host.cuh"

int msm_launch_bn256(const Bases &bs, size_t first, const void *d_scalars, const void *h_scalars, size_t n, const MsmPlan &p, uint64_t *host_windows) {
    return msm_launch<Fq29, FrP>(bs, first, d_scalars, h_scalars, n, p, host_windows);
}
int curve_init_bn256() { return curve_init<Fq29, FrP>(); }
int convert_bases_bn256(const void *d_src, void *d_dst, size_t n) { return convert_bases<Fq29>(d_src, d_dst, n); }
int synth_scalars_bn256(size_t n, uint64_t index0, uint64_t seed, int kind, void *d_out) { return synth_scalars<FrP>(n, index0, seed, kind, d_out); }
int synth_bases_bn256(size_t n, uint64_t index0, uint64_t seed, void *d_out) {
    return synth_bases<FqP>(n, index0, seed, reinterpret_cast<const unsigned char *>(g.consts.p) + 0, d_out);
}
int check_bases_bn256(const Bases &bs, uint32_t *d_bad) {
    return check_bases<Fq29>(bs, reinterpret_cast<const unsigned char *>(g.consts.p) + 128, d_bad);
}
int export_bases_bn256(const Bases &bs, size_t first, size_t n, void *d_out) { return export_bases<Fq29>(bs, first, n, d_out); }
int msm_launch_table_bn256(const Bases &bs, size_t first, const void *d_scalars, size_t n, uint64_t *host_sums) {
    return msm_launch_table<Fq29, FrP>(bs, first, d_scalars, n, host_sums);
}
int build_tables_bn256(Bases &bs, uint32_t c, uint32_t W) { return build_tables<Fq29>(bs, c, W); }
int build_glv_bn256(Bases &bs, const void *d_beta_r261) { return build_glv<Fq29>(bs, d_beta_r261); }
int load_bases_file_bn256(Bases &b, int fd, bool validate, uint32_t *d_bad) {
    return load_bases_file<Fq29>(b, fd, validate, reinterpret_cast<const unsigned char *>(g.consts.p) + 128, d_bad);
}
int save_bases_file_bn256(const Bases &b, int fd) { return save_bases_file<Fq29>(b, fd); }

#ifdef MSM_PROBE_STAMPS
// timing probe only (msm_kernels.cuh): the stamps of the LAST bn256 k_accumulate launch
extern "C" int mira_debug_acc_stamps(uint64_t *out, size_t n_words) {
    if (n_words > 4096 * 3) n_words = 4096 * 3;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_acc_stamps), n_words * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
