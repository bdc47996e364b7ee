// Discrete-gamma rate categories (Yang 1994): replaces corax_compute_gamma_cats
// (called at /root/reference/src/model.cpp:239-270).  R equiprobable
// categories with mean 1; MEAN mode uses the category means (differences of
// the incomplete gamma function at shape+1), MEDIAN mode the category medians
// renormalised.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace {

// regularised lower incomplete gamma P(a, x)
double reg_lower_gamma(double a, double x) {
  if (!(x > 0.0)) return 0.0;
  const double front = std::exp(a * std::log(x) - x - std::lgamma(a));
  if (x < a + 1.0) {
    // power series  x^a e^-x / Gamma(a) * sum_n x^n / (a (a+1) ... (a+n))
    double denom = a, term = 1.0 / a, sum = term;
    for (int n = 1; n < 5000; ++n) {
      denom += 1.0;
      term *= x / denom;
      sum += term;
      if (std::fabs(term) <= std::fabs(sum) * 1e-17) break;
    }
    return front * sum;
  }
  // modified Lentz evaluation of the continued fraction for Q(a, x)
  const double fpmin = 1e-300;
  double b = x + 1.0 - a;
  double c = 1.0 / fpmin, d = 1.0 / b, h = d;
  for (int i = 1; i < 5000; ++i) {
    const double an = -1.0 * i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (std::fabs(d) < fpmin) d = fpmin;
    c = b + an / c;
    if (std::fabs(c) < fpmin) c = fpmin;
    d = 1.0 / d;
    const double delta = d * c;
    h *= delta;
    if (std::fabs(delta - 1.0) <= 1e-17) break;
  }
  return 1.0 - front * h;
}

// x with P(a, x) = prob  (unit rate)
double gamma_quantile_unit(double a, double prob) {
  // bracket, then safeguarded Newton
  double lo = 0.0, hi = std::fmax(1.0, a);
  while (reg_lower_gamma(a, hi) < prob) { lo = hi; hi *= 2.0; }
  double x = 0.5 * (lo + hi);
  const double lg = std::lgamma(a);
  for (int it = 0; it < 300; ++it) {
    const double f = reg_lower_gamma(a, x) - prob;
    if (f < 0.0) lo = x; else hi = x;
    const double pdf = std::exp((a - 1.0) * std::log(x) - x - lg);
    double nx = pdf > 0.0 ? x - f / pdf : 0.5 * (lo + hi);
    if (!(nx > lo && nx < hi)) nx = 0.5 * (lo + hi);
    if (std::fabs(nx - x) <= 1e-16 * std::fabs(nx)) { x = nx; break; }
    x = nx;
  }
  return x;
}

}  // namespace

extern "C" int rdamd_compute_gamma_cats(double alpha, unsigned int categories,
                                        double *output_rates, int rates_mode) {
  rdamd::clear_error();
  if (!(alpha > 0.0) || categories < 1 || !output_rates) {
    rdamd::set_error(30, "rdamd_compute_gamma_cats: invalid alpha (%g) or category count (%u)",
                     alpha, categories);
    return RDAMD_FAILURE;
  }
  if (categories == 1) {
    output_rates[0] = 1.0;
    return RDAMD_SUCCESS;
  }
  const unsigned n = categories;
  const double beta = alpha;   // mean-one gamma
  if (rates_mode == RDAMD_GAMMA_RATES_MEDIAN) {
    double total = 0.0;
    for (unsigned i = 0; i < n; ++i) {
      output_rates[i] = gamma_quantile_unit(alpha, (2.0 * i + 1.0) / (2.0 * n)) / beta;
      total += output_rates[i];
    }
    for (unsigned i = 0; i < n; ++i) output_rates[i] *= n / total;
    return RDAMD_SUCCESS;
  }
  if (rates_mode != RDAMD_GAMMA_RATES_MEAN) {
    rdamd::set_error(31, "rdamd_compute_gamma_cats: unknown mode %d", rates_mode);
    return RDAMD_FAILURE;
  }
  // cumulative mass of the size-biased gamma (shape+1) at the category cut points
  std::vector<double> cum(n + 1, 0.0);
  cum[n] = 1.0;
  for (unsigned i = 1; i < n; ++i)
    cum[i] = reg_lower_gamma(alpha + 1.0, gamma_quantile_unit(alpha, (double)i / n));
  for (unsigned i = 0; i < n; ++i) output_rates[i] = (cum[i + 1] - cum[i]) * n;
  return RDAMD_SUCCESS;
}
