// Host side of the server's pose graph in C++ (header-only, C++14, no dependencies): what coxgraph drives through
// voxgraph + Ceres, over the GPU registration cost of coxgraph_hip_adapters.hpp.
//
//   PoseGraphInterface::addSubmap / addLoopClosureMeasurement / addForceRegistrationConstraint /
//   updateSubmapRPConstraints / optimize(enable_registration)   coxgraph/src/server/pose_graph_interface.cpp:10-105
//   relative-pose residual  r = sqrt_information * e             coxgraph/include/coxgraph/server/backend/relative_pose_constraint.h:28-61,114-119
//   node 0 constant, yaw as an angle                             pose_graph_interface.cpp:20-25, backend/node_collection.h:22-24
//   solver budget: parameter_tolerance 3e-3                      backend/pose_graph.h:60-64
//
// Ceres is replaced by a small dense Levenberg-Marquardt (the graphs have tens of nodes of 4 doubles); registration
// constraints are evaluated in their fused form (8x8 normal equations per constraint), all of them begun before any is
// collected.  The arithmetic is the same as coxgraph_amd/posegraph.py, operation for operation, so the two agree to
// rounding (tests/test_host_logic.py compares them).
#pragma once
#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <set>
#include <vector>

#include "coxgraph_hip_adapters.hpp"

namespace coxgraph_hip {

struct Pose4 {
  double v[4] = {0, 0, 0, 0};  // x, y, z, yaw
};

inline double normalizeAngle(double a) { return a - 2.0 * M_PI * std::floor((a + M_PI) / (2.0 * M_PI)); }

// Cholesky factor L^T of a 4x4 information matrix (relative_pose_constraint.h:28-61: LLT; semi-definite input falls back to
// the root of the eigen-decomposition, like the LDLT branch there)
inline void sqrtInformation(const double info[16], double out[16]) {
  double L[16] = {0};
  bool ok = true;
  for (int j = 0; j < 4 && ok; ++j) {
    double d = info[4 * j + j];
    for (int k = 0; k < j; ++k) d -= L[4 * j + k] * L[4 * j + k];
    if (!(d > 0.0)) {
      ok = false;
      break;
    }
    L[4 * j + j] = std::sqrt(d);
    for (int i = j + 1; i < 4; ++i) {
      double s = info[4 * i + j];
      for (int k = 0; k < j; ++k) s -= L[4 * i + k] * L[4 * j + k];
      L[4 * i + j] = s / L[4 * j + j];
    }
  }
  if (ok) {
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) out[4 * i + j] = L[4 * j + i];  // transpose
    return;
  }
  // cyclic Jacobi eigen-decomposition of the symmetric matrix: info = V diag(w) V^T, root = (V sqrt(max(w, 0)))^T
  double A[16], V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::copy(info, info + 16, A);
  for (int sweep = 0; sweep < 50; ++sweep)
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        if (std::fabs(A[4 * p + q]) < 1e-300) continue;
        const double th = 0.5 * std::atan2(2.0 * A[4 * p + q], A[4 * q + q] - A[4 * p + p]);
        const double c = std::cos(th), s = std::sin(th);
        for (int k = 0; k < 4; ++k) {
          const double akp = A[4 * k + p], akq = A[4 * k + q];
          A[4 * k + p] = c * akp - s * akq;
          A[4 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 4; ++k) {
          const double apk = A[4 * p + k], aqk = A[4 * q + k];
          A[4 * p + k] = c * apk - s * aqk;
          A[4 * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = V[4 * k + p], vkq = V[4 * k + q];
          V[4 * k + p] = c * vkp - s * vkq;
          V[4 * k + q] = s * vkp + c * vkq;
        }
      }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out[4 * i + j] = V[4 * j + i] * std::sqrt(std::max(A[4 * i + i], 0.0));
}

// voxgraph RelativePoseCostFunction <4,4,4>: e = [Rz(yaw_A)^T (t_B - t_A) - t_obs ; wrap(yaw_B - yaw_A - yaw_obs)]
struct RelativePoseConstraint {
  int a = 0, b = 0;
  double obs[4] = {0, 0, 0, 0};
  double sqrt_info[16];
  RelativePoseConstraint(int a_, int b_, const double T_ab[4], const double information[16]) : a(a_), b(b_) {
    std::copy(T_ab, T_ab + 4, obs);
    sqrtInformation(information, sqrt_info);
  }
  // r (4), Ja, Jb (4x4 row-major), already multiplied by the square-root information
  void evaluate(const Pose4& pa, const Pose4& pb, double r[4], double Ja[16], double Jb[16]) const {
    const double c = std::cos(pa.v[3]), s = std::sin(pa.v[3]);
    const double d[3] = {pb.v[0] - pa.v[0], pb.v[1] - pa.v[1], pb.v[2] - pa.v[2]};
    const double e[4] = {c * d[0] + s * d[1] - obs[0], -s * d[0] + c * d[1] - obs[1], d[2] - obs[2], normalizeAngle(pb.v[3] - pa.v[3] - obs[3])};
    const double ja[16] = {-c, -s, 0.0, -s * d[0] + c * d[1], s, -c, 0.0, -c * d[0] - s * d[1], 0.0, 0.0, -1.0, 0.0, 0.0, 0.0, 0.0, -1.0};
    const double jb[16] = {c, s, 0.0, 0.0, -s, c, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0};
    for (int i = 0; i < 4; ++i) {
      r[i] = 0.0;
      for (int k = 0; k < 4; ++k) r[i] += sqrt_info[4 * i + k] * e[k];
      for (int j = 0; j < 4; ++j) {
        double sa = 0.0, sb = 0.0;
        for (int k = 0; k < 4; ++k) {
          sa += sqrt_info[4 * i + k] * ja[4 * k + j];
          sb += sqrt_info[4 * i + k] * jb[4 * k + j];
        }
        Ja[4 * i + j] = sa;
        Jb[4 * i + j] = sb;
      }
    }
  }
};

// voxgraph RegistrationConstraint: reference submap a (registration points) against reading submap b (TSDF layer); one
// RegistrationCostFunction per constraint (a handle has one evaluation in flight)
struct RegistrationConstraint {
  int a = 0, b = 0;
  RegistrationCostFunction* cost = nullptr;  // not owned
};

class PoseGraph {
 public:
  struct Summary {
    double initial_cost = 0.0, final_cost = 0.0;
    int iterations = 0, evaluations = 0;
  };
  std::map<int, Pose4> poses;
  std::set<int> constant;
  std::vector<RelativePoseConstraint> rel, submap_rel;  // submap_rel: consecutive-submap constraints, reset on every update
  std::vector<RegistrationConstraint> reg;

  void addNode(int id, const Pose4& pose, bool is_constant = false) {
    poses[id] = pose;
    if (is_constant) constant.insert(id);
  }

  // cost, gradient g (4F), Gauss-Newton matrix H (4F x 4F row-major) over the F free nodes in id order
  double build(const std::map<int, Pose4>& P, bool exclude_registration, std::vector<double>* g, std::vector<double>* H, std::vector<int>* free_ids) const {
    free_ids->clear();
    std::map<int, int> idx;
    for (const auto& kv : poses)
      if (!constant.count(kv.first)) {
        idx[kv.first] = static_cast<int>(free_ids->size());
        free_ids->push_back(kv.first);
      }
    const int n = 4 * static_cast<int>(free_ids->size());
    g->assign(n, 0.0);
    H->assign(static_cast<size_t>(n) * n, 0.0);
    double cost = 0.0;
    auto scatter = [&](int na, int nb, const double* Haa, const double* Hab, const double* Hbb, const double* ga, const double* gb) {
      const auto ia = idx.find(na), ib = idx.find(nb);
      if (ia != idx.end())
        for (int r = 0; r < 4; ++r) {
          for (int c = 0; c < 4; ++c) (*H)[static_cast<size_t>(4 * ia->second + r) * n + 4 * ia->second + c] += Haa[4 * r + c];
          (*g)[4 * ia->second + r] += ga[r];
        }
      if (ib != idx.end())
        for (int r = 0; r < 4; ++r) {
          for (int c = 0; c < 4; ++c) (*H)[static_cast<size_t>(4 * ib->second + r) * n + 4 * ib->second + c] += Hbb[4 * r + c];
          (*g)[4 * ib->second + r] += gb[r];
        }
      if (ia != idx.end() && ib != idx.end())
        for (int r = 0; r < 4; ++r)
          for (int c = 0; c < 4; ++c) {
            (*H)[static_cast<size_t>(4 * ia->second + r) * n + 4 * ib->second + c] += Hab[4 * r + c];
            (*H)[static_cast<size_t>(4 * ib->second + c) * n + 4 * ia->second + r] += Hab[4 * r + c];
          }
    };
    if (!exclude_registration) {
      // begin every constraint's evaluation, then collect: the kernels overlap, one latency round per evaluation of the graph
      for (const RegistrationConstraint& c : reg)
        if (!c.cost->BeginNormalEquations(P.at(c.a).v, P.at(c.b).v)) throw std::runtime_error("registration constraint: begin failed");
      for (const RegistrationConstraint& c : reg) {
        double H8[64], b8[8], ck = 0.0;
        if (!c.cost->FinishNormalEquations(H8, b8, &ck)) throw std::runtime_error("registration constraint: finish failed");
        double Haa[16], Hab[16], Hbb[16];
        for (int r = 0; r < 4; ++r)
          for (int cc = 0; cc < 4; ++cc) {
            Haa[4 * r + cc] = H8[8 * r + cc];
            Hab[4 * r + cc] = H8[8 * r + 4 + cc];
            Hbb[4 * r + cc] = H8[8 * (r + 4) + 4 + cc];
          }
        scatter(c.a, c.b, Haa, Hab, Hbb, b8, b8 + 4);
        cost += ck;
      }
    }
    for (const auto* list : {&rel, &submap_rel})
      for (const RelativePoseConstraint& c : *list) {
        double r[4], Ja[16], Jb[16], Haa[16], Hab[16], Hbb[16], ga[4], gb[4];
        c.evaluate(P.at(c.a), P.at(c.b), r, Ja, Jb);
        for (int i = 0; i < 4; ++i) {
          ga[i] = gb[i] = 0.0;
          for (int k = 0; k < 4; ++k) {
            ga[i] += Ja[4 * k + i] * r[k];
            gb[i] += Jb[4 * k + i] * r[k];
          }
          for (int j = 0; j < 4; ++j) {
            double aa = 0.0, ab = 0.0, bb = 0.0;
            for (int k = 0; k < 4; ++k) {
              aa += Ja[4 * k + i] * Ja[4 * k + j];
              ab += Ja[4 * k + i] * Jb[4 * k + j];
              bb += Jb[4 * k + i] * Jb[4 * k + j];
            }
            Haa[4 * i + j] = aa;
            Hab[4 * i + j] = ab;
            Hbb[4 * i + j] = bb;
          }
        }
        scatter(c.a, c.b, Haa, Hab, Hbb, ga, gb);
        cost += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
      }
    return cost;
  }

  // Levenberg-Marquardt with the Ceres parameter_tolerance test (backend/pose_graph.h:60)
  Summary optimize(bool exclude_registration, int max_iterations = 50, double parameter_tolerance = 3e-3) {
    std::map<int, Pose4> P = poses;
    std::vector<double> g, H, g2, H2;
    std::vector<int> free_ids;
    double cost = build(P, exclude_registration, &g, &H, &free_ids);
    Summary S;
    S.initial_cost = cost;
    S.evaluations = 1;
    double lam = 1e-4;
    const int n = static_cast<int>(g.size());
    if (n > 0) {
      for (int it = 1; it <= max_iterations; ++it) {
        S.iterations = it;
        std::vector<double> A(H), rhs(n), delta(n);
        for (int i = 0; i < n; ++i) {
          A[static_cast<size_t>(i) * n + i] += lam * std::max(H[static_cast<size_t>(i) * n + i], 1e-12);
          rhs[i] = -g[i];
        }
        if (!solve(&A, &rhs, n, &delta)) {
          lam *= 10.0;
          continue;
        }
        std::map<int, Pose4> trial = P;
        for (size_t k = 0; k < free_ids.size(); ++k) {
          Pose4& t = trial[free_ids[k]];
          for (int c = 0; c < 3; ++c) t.v[c] += delta[4 * k + c];
          t.v[3] = normalizeAngle(t.v[3] + delta[4 * k + 3]);  // angle local parameterisation
        }
        std::vector<int> ids2;
        const double c2 = build(trial, exclude_registration, &g2, &H2, &ids2);
        S.evaluations += 1;
        if (c2 < cost) {
          double x2 = 0.0, d2 = 0.0;
          for (int id : free_ids)
            for (int c = 0; c < 4; ++c) x2 += P[id].v[c] * P[id].v[c];
          for (double d : delta) d2 += d * d;
          P = trial;
          cost = c2;
          g = g2;
          H = H2;
          lam = std::max(lam / 3.0, 1e-12);
          if (std::sqrt(d2) <= parameter_tolerance * (std::sqrt(x2) + parameter_tolerance)) break;
        } else {
          lam *= 4.0;
          if (lam > 1e12) break;
        }
      }
    }
    poses = P;
    S.final_cost = cost;
    return S;
  }

 private:
  // Gaussian elimination with partial pivoting; false when singular
  static bool solve(std::vector<double>* A, std::vector<double>* b, int n, std::vector<double>* x) {
    std::vector<double>& M = *A;
    std::vector<double>& r = *b;
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
    for (int i = n - 1; i >= 0; --i) {
      double s = r[i];
      for (int j = i + 1; j < n; ++j) s -= M[static_cast<size_t>(i) * n + j] * (*x)[j];
      (*x)[i] = s / M[static_cast<size_t>(i) * n + i];
    }
    return true;
  }
};

// The facade coxgraph's server calls (coxgraph/include/coxgraph/server/pose_graph_interface.h:66-90)
class PoseGraphInterface {
 public:
  PoseGraphInterface() {
    // coxgraph/config/server.yaml:37-51
    const double lc[4] = {100.0, 100.0, 250.0, 250.0}, rp[4] = {1000.0, 1000.0, 2500.0, 2500.0};
    std::fill(lc_info_, lc_info_ + 16, 0.0);
    std::fill(sm_rp_info_, sm_rp_info_ + 16, 0.0);
    for (int i = 0; i < 4; ++i) {
      lc_info_[5 * i] = lc[i];
      sm_rp_info_[5 * i] = rp[i];
    }
  }
  // pose_graph_interface.cpp:10-30: submap 0 is constant
  void addSubmap(int submap_id, const Pose4& pose, int client_id = 0) {
    clients_[submap_id] = client_id;
    pose_graph_.addNode(submap_id, pose, submap_id == 0);
  }
  bool addLoopClosureMeasurement(int a, int b, const double T_ab[4]) {
    pose_graph_.rel.emplace_back(a, b, T_ab, lc_info_);
    return true;
  }
  // pose_graph_interface.cpp:88-105
  void addForceRegistrationConstraint(int a, int b, RegistrationCostFunction* cost) { pose_graph_.reg.push_back(RegistrationConstraint{a, b, cost}); }
  // pose_graph_interface.cpp:51-71: consecutive submaps of one client keep their current relative pose
  void updateSubmapRPConstraints() {
    pose_graph_.submap_rel.clear();
    std::map<int, std::vector<int>> by_client;
    for (const auto& kv : clients_) by_client[kv.second].push_back(kv.first);  // std::map: ids ascending
    for (const auto& kv : by_client)
      for (size_t k = 0; k + 1 < kv.second.size(); ++k) {
        const int i = kv.second[k], j = kv.second[k + 1];
        const Pose4 &pa = pose_graph_.poses[i], &pb = pose_graph_.poses[j];
        const double c = std::cos(pa.v[3]), s = std::sin(pa.v[3]);
        const double d[3] = {pb.v[0] - pa.v[0], pb.v[1] - pa.v[1], pb.v[2] - pa.v[2]};
        const double T_ij[4] = {c * d[0] + s * d[1], -s * d[0] + c * d[1], d[2], normalizeAngle(pb.v[3] - pa.v[3])};
        pose_graph_.submap_rel.emplace_back(i, j, T_ij, sm_rp_info_);
      }
  }
  // pose_graph_interface.cpp:32-49: first without registration constraints, then with all constraints
  std::pair<PoseGraph::Summary, PoseGraph::Summary> optimize(bool enable_registration = true) {
    const PoseGraph::Summary first = pose_graph_.optimize(true);
    const PoseGraph::Summary second = pose_graph_.optimize(!enable_registration);
    return {first, second};
  }
  const std::map<int, Pose4>& getPoseMap() const { return pose_graph_.poses; }
  PoseGraph& poseGraph() { return pose_graph_; }

 private:
  PoseGraph pose_graph_;
  std::map<int, int> clients_;
  double lc_info_[16], sm_rp_info_[16];
};

}  // namespace coxgraph_hip
