// grumpkin: instantiates the MSM pipeline for this curve (coordinates Fr29 / FrP, scalars FqP).
#include "msm_host.cuh"

int msm_launch_grumpkin(const Bases &bs, size_t first, const void *d_scalars, const void *h_scalars, size_t n, const MsmPlan &p, uint64_t *host_windows) {
    return msm_launch<Fr29, FqP>(bs, first, d_scalars, h_scalars, n, p, host_windows);
}
int curve_init_grumpkin() { return curve_init<Fr29, FqP>(); }
int convert_bases_grumpkin(const void *d_src, void *d_dst, size_t n) { return convert_bases<Fr29>(d_src, d_dst, n); }
int synth_scalars_grumpkin(size_t n, uint64_t index0, uint64_t seed, int kind, void *d_out) { return synth_scalars<FqP>(n, index0, seed, kind, d_out); }
int synth_bases_grumpkin(size_t n, uint64_t index0, uint64_t seed, void *d_out) {
    return synth_bases<FrP>(n, index0, seed, reinterpret_cast<const unsigned char *>(g.consts.p) + 64, d_out);
}
int check_bases_grumpkin(const Bases &bs, uint32_t *d_bad) {
    return check_bases<Fr29>(bs, reinterpret_cast<const unsigned char *>(g.consts.p) + 160, d_bad);
}
int export_bases_grumpkin(const Bases &bs, size_t first, size_t n, void *d_out) { return export_bases<Fr29>(bs, first, n, d_out); }
int msm_launch_table_grumpkin(const Bases &bs, size_t first, const void *d_scalars, size_t n, uint64_t *host_sums) {
    return msm_launch_table<Fr29, FqP>(bs, first, d_scalars, n, host_sums);
}
int build_tables_grumpkin(Bases &bs, uint32_t c, uint32_t W) { return build_tables<Fr29>(bs, c, W); }
int build_glv_grumpkin(Bases &bs, const void *d_beta_r261) { return build_glv<Fr29>(bs, d_beta_r261); }
int load_bases_file_grumpkin(Bases &b, int fd, bool validate, uint32_t *d_bad) {
    return load_bases_file<Fr29>(b, fd, validate, reinterpret_cast<const unsigned char *>(g.consts.p) + 160, d_bad);
}
int save_bases_file_grumpkin(const Bases &b, int fd) { return save_bases_file<Fr29>(b, fd); }
