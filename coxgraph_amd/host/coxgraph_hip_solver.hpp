// ceres::Solve as coxgraph's server runs it ([R] coxgraph/include/coxgraph/server/backend/pose_graph.h:56-68: parameter_tolerance 3e-3,
// max_solver_time_in_seconds 4, every other Solver::Options field at Ceres' default), restated from Ceres' published description
// of its trust-region minimiser because Ceres itself is not available here: TrustRegionMinimizer with the LEVENBERG_MARQUARDT
// strategy, monotonic steps, Jacobi scaling, on dense normal equations (the graphs have tens of nodes of 4 doubles; Ceres'
// SPARSE_SCHUR solves the same linear system).  Header-only, C++14.  The same policy, operation for operation, as
// coxgraph_amd/posegraph.py::trust_region_minimize (tests/test_host_logic.py compares the two and checks both against an iteration
// trace derived by hand).
//
// Per iteration:
//   step      D^2 = clamp(diag(H_s), min_lm_diagonal, max_lm_diagonal) / radius;  (H_s + D^2) d_s = -g_s;  d = scale * d_s
//             (_s: Jacobian columns scaled by 1 / (1 + ||column||), norms taken once at the initial point: jacobi_scaling)
//   validity  model_cost_change = -d_s . (g_s + H_s d_s / 2) must be > 0; an invalid step halves the radius, and
//             max_num_consecutive_invalid_steps of them in a row fail the solve
//   stop      ||d|| <= parameter_tolerance (||x|| + parameter_tolerance), then |cost change| <= function_tolerance * cost -- both
//             BEFORE the step is taken, so x stays where it was
//   accept    rho = cost change / model_cost_change > min_relative_decrease: x <- candidate; stop if max |gradient| <=
//             gradient_tolerance; radius <- min(max_radius, radius / max(1/3, 1 - (2 rho - 1)^3)); decrease factor <- 2
//   reject    radius /= decrease factor; decrease factor *= 2
//   before every iteration: max_num_iterations, max_solver_time_in_seconds, radius <= min_trust_region_radius
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <string>
#include <vector>

namespace coxgraph_hip {

struct TrustRegionOptions {
  int max_num_iterations = 50;
  double max_solver_time_in_seconds = 4.0;  // pose_graph.h:61
  double function_tolerance = 1e-6;
  double gradient_tolerance = 1e-10;
  double parameter_tolerance = 3e-3;  // pose_graph.h:60
  double min_relative_decrease = 1e-3;
  double initial_trust_region_radius = 1e4;
  double max_trust_region_radius = 1e16;
  double min_trust_region_radius = 1e-32;
  double min_lm_diagonal = 1e-6;
  double max_lm_diagonal = 1e32;
  int max_num_consecutive_invalid_steps = 5;
  bool jacobi_scaling = true;
};

struct TrustRegionIteration {
  int iteration = 0;
  double cost = 0.0, cost_change = 0.0, gradient_max_norm = 0.0, step_norm = 0.0, relative_decrease = 0.0, trust_region_radius = 0.0;
  bool step_is_valid = false, step_is_successful = false;
};

struct TrustRegionSummary {
  double initial_cost = 0.0, final_cost = 0.0;
  int iterations = 0, evaluations = 0, successful_steps = 0, unsuccessful_steps = 0;
  std::string termination = "CONVERGENCE";  // CONVERGENCE | NO_CONVERGENCE | FAILURE
  std::string message;
  std::vector<TrustRegionIteration> trace;
};

namespace detail {
// Gaussian elimination with partial pivoting; false when singular
inline bool solveDense(std::vector<double> M, std::vector<double> r, int n, std::vector<double>* x) {
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(M[static_cast<size_t>(i) * n + k]) > std::fabs(M[static_cast<size_t>(piv) * n + k])) piv = i;
    if (!(std::fabs(M[static_cast<size_t>(piv) * n + k]) > 1e-300)) return false;
    if (piv != k) {
      for (int j = 0; j < n; ++j) std::swap(M[static_cast<size_t>(k) * n + j], M[static_cast<size_t>(piv) * n + j]);
      std::swap(r[k], r[piv]);
    }
    for (int i = k + 1; i < n; ++i) {
      const double f = M[static_cast<size_t>(i) * n + k] / M[static_cast<size_t>(k) * n + k];
      if (f == 0.0) continue;
      for (int j = k; j < n; ++j) M[static_cast<size_t>(i) * n + j] -= f * M[static_cast<size_t>(k) * n + j];
      r[i] -= f * r[k];
    }
  }
  x->assign(n, 0.0);
  for (int i = n - 1; i >= 0; --i) {
    double s = r[i];
    for (int j = i + 1; j < n; ++j) s -= M[static_cast<size_t>(i) * n + j] * (*x)[j];
    (*x)[i] = s / M[static_cast<size_t>(i) * n + i];
  }
  for (double v : *x)
    if (!std::isfinite(v)) return false;
  return true;
}
inline double norm2(const std::vector<double>& v) {
  double s = 0.0;
  for (double a : v) s += a * a;
  return std::sqrt(s);
}
}  // namespace detail

// evaluate(x, &g, &H) -> cost, with g = J^T f (n) and H = J^T J (n x n row-major); plus(x, delta) -> x (+) delta, the (local)
// parameterisation's Plus.  x is updated in place.
template <typename Evaluate, typename Plus>
TrustRegionSummary trustRegionMinimize(Evaluate evaluate, Plus plus, std::vector<double>* x_io, const TrustRegionOptions& opt = TrustRegionOptions()) {
  const auto t_start = std::chrono::steady_clock::now();
  std::vector<double>& x = *x_io;
  const int n = static_cast<int>(x.size());
  std::vector<double> g, H;
  double cost = evaluate(x, &g, &H);
  TrustRegionSummary S;
  S.initial_cost = S.final_cost = cost;
  S.evaluations = 1;
  auto finish = [&](const char* term, const char* msg) {
    S.final_cost = cost;
    S.termination = term;
    S.message = msg;
    return S;
  };
  if (n == 0) return finish("CONVERGENCE", "no free parameters");
  std::vector<double> scale(n, 1.0);
  if (opt.jacobi_scaling)
    for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(std::max(H[static_cast<size_t>(i) * n + i], 0.0)));
  auto gradientMaxNorm = [&](const std::vector<double>& xx, const std::vector<double>& gg) {
    std::vector<double> neg(n);
    for (int i = 0; i < n; ++i) neg[i] = -gg[i];
    const std::vector<double> y = plus(xx, neg);
    double m = 0.0;
    for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(xx[i] - y[i]));
    return m;
  };
  double gmax = gradientMaxNorm(x, g);
  {
    TrustRegionIteration it;
    it.cost = cost;
    it.gradient_max_norm = gmax;
    it.trust_region_radius = opt.initial_trust_region_radius;
    S.trace.push_back(it);
  }
  if (gmax <= opt.gradient_tolerance) return finish("CONVERGENCE", "Gradient tolerance reached.");
  double radius = opt.initial_trust_region_radius, decrease_factor = 2.0, x_norm = detail::norm2(x);
  int invalid = 0, iteration = 0;
  for (;;) {
    if (iteration >= opt.max_num_iterations) return finish("NO_CONVERGENCE", "Maximum number of iterations reached.");
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() >= opt.max_solver_time_in_seconds)
      return finish("NO_CONVERGENCE", "Maximum solver time reached.");
    if (radius <= opt.min_trust_region_radius) return finish("CONVERGENCE", "Minimum trust region radius reached.");
    ++iteration;
    S.iterations = iteration;
    std::vector<double> Hs(static_cast<size_t>(n) * n), gs(n), A, rhs(n), ds;
    for (int i = 0; i < n; ++i) {
      gs[i] = g[i] * scale[i];
      rhs[i] = -gs[i];
      for (int j = 0; j < n; ++j) Hs[static_cast<size_t>(i) * n + j] = H[static_cast<size_t>(i) * n + j] * scale[i] * scale[j];
    }
    A = Hs;
    for (int i = 0; i < n; ++i) A[static_cast<size_t>(i) * n + i] += std::min(std::max(Hs[static_cast<size_t>(i) * n + i], opt.min_lm_diagonal), opt.max_lm_diagonal) / radius;
    S.trace.emplace_back();
    const size_t ti = S.trace.size() - 1;
    S.trace[ti].iteration = iteration;
    S.trace[ti].cost = cost;
    S.trace[ti].gradient_max_norm = gmax;
    S.trace[ti].trust_region_radius = radius;
    bool valid = detail::solveDense(A, rhs, n, &ds);
    double model_cost_change = 0.0;
    if (valid) {
      double acc = 0.0;
      for (int i = 0; i < n; ++i) {
        double hd = 0.0;
        for (int j = 0; j < n; ++j) hd += Hs[static_cast<size_t>(i) * n + j] * ds[j];
        acc += ds[i] * (gs[i] + 0.5 * hd);
      }
      model_cost_change = -acc;
      valid = model_cost_change > 0.0;
    }
    if (!valid) {
      if (++invalid >= opt.max_num_consecutive_invalid_steps)
        return finish("FAILURE", "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps.");
      radius *= 0.5;
      S.trace[ti].trust_region_radius = radius;
      continue;
    }
    invalid = 0;
    S.trace[ti].step_is_valid = true;
    std::vector<double> delta(n);
    for (int i = 0; i < n; ++i) delta[i] = ds[i] * scale[i];
    const std::vector<double> cand = plus(x, delta);
    std::vector<double> g2, H2;
    const double c2 = evaluate(cand, &g2, &H2);
    S.evaluations += 1;
    double sn = 0.0;
    for (int i = 0; i < n; ++i) sn += (x[i] - cand[i]) * (x[i] - cand[i]);
    const double step_norm = std::sqrt(sn);
    S.trace[ti].step_norm = step_norm;
    S.trace[ti].cost_change = cost - c2;
    if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) return finish("CONVERGENCE", "Parameter tolerance reached.");
    if (std::fabs(cost - c2) <= opt.function_tolerance * cost) return finish("CONVERGENCE", "Function tolerance reached.");
    const double rho = (cost - c2) / model_cost_change;
    S.trace[ti].relative_decrease = rho;
    if (rho > opt.min_relative_decrease) {
      x = cand;
      cost = c2;
      g.swap(g2);
      H.swap(H2);
      x_norm = detail::norm2(x);
      S.trace[ti].step_is_successful = true;
      S.trace[ti].cost = cost;
      S.successful_steps += 1;
      gmax = gradientMaxNorm(x, g);
      S.trace[ti].gradient_max_norm = gmax;
      if (gmax <= opt.gradient_tolerance) return finish("CONVERGENCE", "Gradient tolerance reached.");
      const double t = 2.0 * rho - 1.0;
      radius = std::min(opt.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - t * t * t));
      decrease_factor = 2.0;
    } else {
      S.unsuccessful_steps += 1;
      radius /= decrease_factor;
      decrease_factor *= 2.0;
    }
    S.trace[ti].trust_region_radius = radius;
  }
}

}  // namespace coxgraph_hip
