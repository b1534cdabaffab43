// kmanip_dispatch.hip -- host-side choice of the k_step / k_reset variant (link-count class x solver).
#include "kmanip_device.hpp"

#define KM_DECL(NL, G, S)                                                                                          \
  void kmanip_launch_step_##NL##_##G##_##S(const KDeviceModel*, const KDeviceState&, const float*, double*, double*, uint8_t*, int, hipStream_t); \
  void kmanip_launch_reset_##NL##_##G##_##S(const KDeviceModel*, const KDeviceState&, const uint8_t*, double*, hipStream_t);
KM_DECL(10, 16, 0) KM_DECL(10, 16, 1) KM_DECL(20, 32, 0) KM_DECL(20, 32, 1)
#undef KM_DECL
void kmanip_launch_observe_10_16_1(const KDeviceModel*, const KDeviceState&, double*, double*, hipStream_t);
void kmanip_launch_prepare_10_16_1(KDeviceModel*, hipStream_t);
void kmanip_launch_prepare_20_32_1(KDeviceModel*, hipStream_t);
void kmanip_launch_observe_20_32_1(const KDeviceModel*, const KDeviceState&, double*, double*, hipStream_t);

void kmanip_launch_step(const KDeviceModel* dm, const KModelDesc& hd, const KDeviceState& st, const float* act, double* obs, double* reward,
                        uint8_t* done, int nchunk, hipStream_t stream) {
  const bool newton = hd.solver == KM_SOLVER_NEWTON;
  if (hd.nlink <= 10) { if (newton) kmanip_launch_step_10_16_1(dm, st, act, obs, reward, done, nchunk, stream); else kmanip_launch_step_10_16_0(dm, st, act, obs, reward, done, nchunk, stream); }
  else { if (newton) kmanip_launch_step_20_32_1(dm, st, act, obs, reward, done, nchunk, stream); else kmanip_launch_step_20_32_0(dm, st, act, obs, reward, done, nchunk, stream); }
}
void kmanip_launch_reset(const KDeviceModel* dm, const KModelDesc& hd, const KDeviceState& st, const uint8_t* mask,
                         int use_done_bits, double* obs, hipStream_t stream) {
  (void)use_done_bits;
  const bool newton = hd.solver == KM_SOLVER_NEWTON;
  if (hd.nlink <= 10) { if (newton) kmanip_launch_reset_10_16_1(dm, st, mask, obs, stream); else kmanip_launch_reset_10_16_0(dm, st, mask, obs, stream); }
  else { if (newton) kmanip_launch_reset_20_32_1(dm, st, mask, obs, stream); else kmanip_launch_reset_20_32_0(dm, st, mask, obs, stream); }
}
void kmanip_launch_observe(const KDeviceModel* dm, const KModelDesc& hd, const KDeviceState& st, double* obs, double* reward,
                           hipStream_t stream) {
  if (hd.nlink <= 10) kmanip_launch_observe_10_16_1(dm, st, obs, reward, stream);
  else kmanip_launch_observe_20_32_1(dm, st, obs, reward, stream);
}
void kmanip_launch_prepare_model(KDeviceModel* dm, const KModelDesc& hd, hipStream_t stream) {
  if (hd.nlink <= 10) kmanip_launch_prepare_10_16_1(dm, stream);
  else kmanip_launch_prepare_20_32_1(dm, stream);
}
