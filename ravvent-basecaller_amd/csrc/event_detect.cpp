// Host-side streaming event detector: two-window t-statistic peak detector
// restated from /root/reference/event_detection/event_detector.py:75-210 (EventDetector.run,
// _add_sample, _compute_tstat, _detect_peak, _create_event), itself Scrappie-derived.  The
// reference is a pure-Python per-sample loop (~1e5 samples/s); this is the same arithmetic in
// C++ (float64 ring buffer of running sums, uint32 wrap of the sample clock as in to_u32 :281-283).
// Pinned against the reference's own output: tests/golden/events_*.npz, tests/test_events.py.
#include "../../include/ravvent_hip.h"

#include <cmath>
#include <cstdint>
#include <vector>

namespace {

constexpr double kFltMin = 1.17549435e-38, kFltMax = 3.40282347e+38;   // event_detector.py:10-11

struct Detector {
  double threshold; int window_length;
  uint32_t masked_to = 0; int64_t peak_pos = -1; double peak_value = kFltMax; bool valid_peak = false;
};

struct State {
  int w1, w2, buf_len; double peak_height;
  std::vector<double> sum, sumsq;
  uint32_t t = 1, buf_mid = 0;
  int64_t evt_st = 0; double evt_st_sum = 0., evt_st_sumsq = 0.;
  Detector sd, ld;

  double tstat(int w) const {                                   // _compute_tstat :109-147
    if ((int64_t)t <= 2 * (int64_t)w || w < 2) return 0.;
    const double wf = (double)w;
    const uint32_t i = buf_mid % buf_len, st = (uint32_t)(buf_mid - (uint32_t)w) % buf_len,
                   en = (uint32_t)(buf_mid + (uint32_t)w) % buf_len;
    const double sum1 = sum[i] - sum[st], sumsq1 = sumsq[i] - sumsq[st];
    const double sum2 = sum[en] - sum[i], sumsq2 = sumsq[en] - sumsq[i];
    const double mean1 = sum1 / wf, mean2 = sum2 / wf;
    double var = sumsq1 / wf - mean1 * mean1 + sumsq2 / wf - mean2 * mean2;
    var = std::fmax(var, kFltMin);
    return std::fabs(mean2 - mean1) / std::sqrt(var / wf);
  }

  bool detect_peak(double cur, Detector& d, bool is_short) {    // _detect_peak :149-187
    if (d.masked_to >= buf_mid) return false;
    if (d.peak_pos == -1) {
      if (cur < d.peak_value) d.peak_value = cur;
      else if (cur - d.peak_value > peak_height) { d.peak_value = cur; d.peak_pos = (int32_t)buf_mid; }
    } else {
      if (cur > d.peak_value) { d.peak_value = cur; d.peak_pos = (int32_t)buf_mid; }
      if (is_short && d.peak_value > d.threshold) {
        ld.masked_to = (uint32_t)(d.peak_pos + d.window_length);
        ld.peak_pos = -1; ld.peak_value = kFltMax; ld.valid_peak = false;
      }
      if (d.peak_value - cur > peak_height && d.peak_value > d.threshold) d.valid_peak = true;
      if (d.valid_peak && (double)((int64_t)buf_mid - d.peak_pos) > d.window_length / 2.0) {
        d.peak_pos = -1; d.peak_value = cur; d.valid_peak = false;
        return true;
      }
    }
    return false;
  }
};

}  // namespace

extern "C" int rv_detect_events(const double* raw, size_t n, int32_t w1, int32_t w2, double threshold1,
                                double threshold2, double peak_height, int64_t* start, int64_t* length,
                                double* mean, double* stdv, size_t capacity, size_t* n_events) {
  if (!raw || !n_events || w1 < 1 || w2 < 1 || w2 < w1) return RV_EINVAL;
  State s;
  s.w1 = w1; s.w2 = w2; s.buf_len = 1 + 2 * w2; s.peak_height = peak_height;
  s.sum.assign(s.buf_len, 0.); s.sumsq.assign(s.buf_len, 0.);
  s.sd.threshold = threshold1; s.sd.window_length = w1;
  s.ld.threshold = threshold2; s.ld.window_length = w2;
  // is_short follows the reference's test `detector['window_length'] == short_detector['window_length']`
  // (:168), which is also true for the long detector when both windows are equal.
  const bool long_is_short = w1 == w2;
  size_t cnt = 0;
  for (size_t k = 0; k < n; ++k) {                              // _add_sample :85-107
    const double x = raw[k];
    const uint32_t tm = s.t % s.buf_len;
    const uint32_t prev = tm > 0 ? tm - 1 : s.buf_len - 1;
    s.sum[tm] = s.sum[prev] + x;
    s.sumsq[tm] = s.sumsq[prev] + x * x;
    s.t += 1;
    s.buf_mid = (uint32_t)(s.t - (uint32_t)(s.buf_len / 2) - 1u);
    const double t1 = s.tstat(w1), t2 = s.tstat(w2);
    const bool p1 = s.detect_peak(t1, s.sd, true);
    const bool p2 = s.detect_peak(t2, s.ld, long_is_short);
    if (p1 || p2) {                                             // _create_event :189-210
      const uint32_t evt_en = (uint32_t)(s.buf_mid - (uint32_t)w1 + 1u);
      const uint32_t eb = evt_en % s.buf_len;
      const double len = (double)((int64_t)evt_en - s.evt_st);
      if (len < kFltMin) continue;
      const double m = (s.sum[eb] - s.evt_st_sum) / len;
      double v = (s.sumsq[eb] - s.evt_st_sumsq) / len - m * m;
      v = std::sqrt(std::fmax(v, kFltMin));
      if (cnt < capacity && start && length && mean && stdv) {
        start[cnt] = s.evt_st; length[cnt] = (int64_t)len; mean[cnt] = m; stdv[cnt] = v;
      }
      ++cnt;
      s.evt_st = evt_en; s.evt_st_sum = s.sum[eb]; s.evt_st_sumsq = s.sumsq[eb];
    }
  }
  *n_events = cnt;
  return cnt > capacity && start ? RV_EINVAL : RV_OK;
}
