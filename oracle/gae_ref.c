/* TEST INFRASTRUCTURE ONLY — plain-C restatement of the reference's lambda-return scan and
 * advantage normalisation, used as an independent checker of oracle/numpy_port.py and as the
 * single-core C baseline for the GAE kernel.  Restates tonic/replays/utils.py:4-19 and
 * tonic/replays/segments.py:41-46 (paths relative to the reference checkout).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no FMA, same roundings as NumPy).
 */
#include <math.h>
#include <stdint.h>

/* [T,W] row-major float32; returns[t] = rewards[t] + gamma * bootstrap, scanned from T-1. */
void oracle_lambda_returns(const float* next_values, const float* rewards, const float* resets,
                           const float* terminations, float* returns, int64_t T, int64_t W,
                           double discount_factor, double trace_decay) {
  const float gamma = (float)discount_factor, lambda = (float)trace_decay;
  const float one_minus_lambda = (float)(1.0 - trace_decay);   /* utils.py:14: f64 then f32 */
  for (int64_t w = 0; w < W; ++w) {
    float last = next_values[(T - 1) * W + w];                 /* utils.py:11 */
    for (int64_t t = T - 1; t >= 0; --t) {
      const int64_t i = t * W + w;
      float boot = one_minus_lambda * next_values[i] + lambda * last;
      boot = boot * (1.0f - resets[i]);
      boot = boot + resets[i] * next_values[i];
      boot = boot * (1.0f - terminations[i]);
      last = rewards[i] + gamma * boot;
      returns[i] = last;
    }
  }
}

/* raw advantages + their float64 mean / population std (segments.py:42-45). */
void oracle_advantage_stats(const float* returns, const float* values, float* advantages,
                            double* mean_std, int64_t n) {
  double sum = 0.0, sum_sq = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const float a = returns[i] - values[i];
    advantages[i] = a;
    sum += a;
    sum_sq += (double)a * a;
  }
  const double mean = sum / (double)n;
  double var = sum_sq / (double)n - mean * mean;
  if (var < 0) var = 0;
  mean_std[0] = mean;
  mean_std[1] = sqrt(var);
}
