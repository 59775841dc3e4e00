// Host side of the server's pose graph in C++ (header-only, C++14, no dependencies): what coxgraph drives through
// voxgraph + Ceres, over the GPU registration cost of coxgraph_hip_adapters.hpp.
//
//   PoseGraphInterface::addSubmap / addLoopClosureMeasurement / addForceRegistrationConstraint /
//   updateSubmapRPConstraints / optimize(enable_registration)   coxgraph/src/server/pose_graph_interface.cpp:10-105
//   relative-pose residual  r = sqrt_information * e             coxgraph/include/coxgraph/server/backend/relative_pose_constraint.h:28-61,114-119
//   node 0 constant, yaw as an angle                             pose_graph_interface.cpp:20-25, backend/node_collection.h:22-24
//   ceres::Solve, parameter_tolerance 3e-3, 4 s budget           backend/pose_graph.h:56-68
//
// Ceres itself is not in the image: its trust-region minimiser is restated in coxgraph_hip_solver.hpp (Solver::Options defaults +
// the two values coxgraph overrides) on dense normal equations (the graphs have tens of nodes of 4 doubles); registration
// constraints are evaluated in their fused form (8x8 normal equations per constraint), all of them begun before any is
// collected.  The arithmetic is the same as coxgraph_amd/posegraph.py, operation for operation, so the two agree to
// rounding (tests/test_host_logic.py compares them).
#pragma once
#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include <cstdio>

#include "coxgraph_hip_solver.hpp"
#include "coxgraph_hip_submap.hpp"

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
  RegistrationCostFunction* cost = nullptr;          // the evaluator (== owned.get() when the pose graph built it itself)
  std::shared_ptr<RegistrationCostFunction> owned;   // constraints created from the submap collection
  double sampling_ratio = -1.0;                      // < 0: every point once; else int(ratio * |set|) weighted draws per solve
};

class PoseGraph {
 public:
  typedef TrustRegionSummary Summary;
  TrustRegionOptions solver_options;  // ceres::Solver::Options as backend/pose_graph.h:56-68 sets them
  std::map<int, Pose4> poses;
  std::set<int> constant;
  std::vector<RelativePoseConstraint> rel, submap_rel;  // submap_rel: consecutive-submap constraints, reset on every update
  std::vector<RegistrationConstraint> reg;        // forced registration constraints (addForceRegistrationConstraint): kept
  std::vector<RegistrationConstraint> overlap_reg;  // overlap-driven ones: rebuilt by updateRegistrationConstraints()
  uint64_t solve_counter = 0;                     // seeds the weighted sampler: fresh, reproducible draws per solve
  // multi-GPU server: registration constraint k is evaluated by rank k % world only; every evaluation sums ONE packed
  // buffer ((4F)^2 + 4F + 1 doubles) over the ranks with RCCL (cox_comm_allreduce_f64), then all ranks solve the same
  // small system redundantly -- no broadcast needed.  Every rank holds the whole graph (constraints it does not own need
  // no evaluator: cost == nullptr).
  cox_comm_t* comm = nullptr;

  void resetRegistrationConstraints() { overlap_reg.clear(); }
  void resetSubmapRelativePoseConstraints() { submap_rel.clear(); }

  // residual vectors of one class of relative-pose constraints at the current poses (4 numbers per constraint)
  std::vector<double> evaluateResiduals(const std::vector<RelativePoseConstraint>& list) const {
    std::vector<double> out;
    for (const RelativePoseConstraint& c : list) {
      double r[4], Ja[16], Jb[16];
      c.evaluate(poses.at(c.a), poses.at(c.b), r, Ja, Jb);
      out.insert(out.end(), r, r + 4);
    }
    return out;
  }

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
      int rank = 0, world = 1;
      if (comm) check(cox_comm_rank(comm, &rank, &world), "comm rank");
      std::vector<const RegistrationConstraint*> all;
      size_t k_all = 0;
      for (const auto* list : {&reg, &overlap_reg})
        for (const RegistrationConstraint& c : *list)
          if (static_cast<int>(k_all++ % static_cast<size_t>(world)) == rank) all.push_back(&c);
      std::vector<double> g_keep, H_keep;
      double cost_keep = 0.0;
      if (comm) {  // accumulate this rank's share on its own, reduce it, then add the (redundantly evaluated) rest
        g_keep.assign(n, 0.0);
        H_keep.assign(static_cast<size_t>(n) * n, 0.0);
        g_keep.swap(*g);
        H_keep.swap(*H);
        cost_keep = cost;
        cost = 0.0;
      }
      // A rank whose evaluation fails must still take part in the all-reduce -- RCCL has no timeout, the other ranks would wait in
      // ncclAllReduce for ever: it contributes zeros and one in the buffer's extra "failed ranks" word, and every rank throws.
      std::string local_error;
      try {
        // ONE launch for all of this rank's constraints (cox_reg_normal_eq_batch); a single constraint takes the call of its own
        std::vector<double> H8s(64 * all.size()), b8s(8 * all.size()), cks(all.size(), 0.0);
        if (all.size() >= 2) {
          std::vector<const RegistrationCostFunction*> costs;
          std::vector<double> pr, pd;
          for (const RegistrationConstraint* cp : all) {
            costs.push_back(cp->cost);
            pr.insert(pr.end(), P.at(cp->a).v, P.at(cp->a).v + 4);
            pd.insert(pd.end(), P.at(cp->b).v, P.at(cp->b).v + 4);
          }
          if (!RegistrationCostFunction::NormalEquationsBatch(costs, pr.data(), pd.data(), H8s.data(), b8s.data(), cks.data()))
            throw std::runtime_error("registration constraints: batched evaluation failed");
        } else {
          for (size_t k = 0; k < all.size(); ++k)
            if (!all[k]->cost->NormalEquations(P.at(all[k]->a).v, P.at(all[k]->b).v, &H8s[64 * k], &b8s[8 * k], &cks[k]))
              throw std::runtime_error("registration constraint: evaluation failed");
        }
        for (size_t k = 0; k < all.size(); ++k) {
          const RegistrationConstraint& c = *all[k];
          const double *H8 = &H8s[64 * k], *b8 = &b8s[8 * k];
          double Haa[16], Hab[16], Hbb[16];
          for (int r = 0; r < 4; ++r)
            for (int cc = 0; cc < 4; ++cc) {
              Haa[4 * r + cc] = H8[8 * r + cc];
              Hab[4 * r + cc] = H8[8 * r + 4 + cc];
              Hbb[4 * r + cc] = H8[8 * (r + 4) + 4 + cc];
            }
          scatter(c.a, c.b, Haa, Hab, Hbb, b8, b8 + 4);
          cost += cks[k];
        }
      } catch (const std::exception& e) {
        if (!comm) throw;
        local_error = e.what();
        std::fill(g->begin(), g->end(), 0.0);
        std::fill(H->begin(), H->end(), 0.0);
        cost = 0.0;
      }
      if (comm) {
        std::vector<double> buf(static_cast<size_t>(n) * n + n + 2);
        std::copy(H->begin(), H->end(), buf.begin());
        std::copy(g->begin(), g->end(), buf.begin() + static_cast<size_t>(n) * n);
        buf[static_cast<size_t>(n) * n + n] = cost;
        buf.back() = local_error.empty() ? 0.0 : 1.0;
        check(cox_comm_allreduce_f64(comm, buf.data(), buf.size()), "all-reduce of the normal equations");
        if (buf.back() != 0.0)
          throw std::runtime_error("registration constraints: evaluation failed on " + std::to_string(static_cast<int>(buf.back())) + " rank(s)" +
                                   (local_error.empty() ? std::string() : ": " + local_error));
        for (size_t i = 0; i < H->size(); ++i) (*H)[i] = H_keep[i] + buf[i];
        for (int i = 0; i < n; ++i) (*g)[i] = g_keep[i] + buf[static_cast<size_t>(n) * n + i];
        cost = cost_keep + buf[static_cast<size_t>(n) * n + n];
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

  // ceres::Solve(options, &problem, &summary) of backend/pose_graph.h:56-73 (the policy: coxgraph_hip_solver.hpp)
  Summary optimize(bool exclude_registration = false) {
    if (!exclude_registration) {  // the sampler's draws of this solve (voxgraph redraws in every Evaluate; one set per solve keeps the cost comparisons of the trust region meaningful)
      ++solve_counter;
      uint64_t k = 0;
      for (auto* list : {&reg, &overlap_reg})
        for (RegistrationConstraint& c : *list) {
          ++k;
          if (c.cost && c.sampling_ratio >= 0.0) c.cost->drawSamples(static_cast<uint64_t>(c.sampling_ratio * static_cast<double>(c.cost->num_points())), solve_counter * 1000003ull + k);
        }
    }
    std::vector<int> free_ids;
    for (const auto& kv : poses)
      if (!constant.count(kv.first)) free_ids.push_back(kv.first);
    std::vector<double> x;
    for (int id : free_ids) x.insert(x.end(), poses[id].v, poses[id].v + 4);
    auto unpack = [&](const std::vector<double>& xx) {
      std::map<int, Pose4> P = poses;
      for (size_t k = 0; k < free_ids.size(); ++k)
        for (int c = 0; c < 4; ++c) P[free_ids[k]].v[c] = xx[4 * k + c];
      return P;
    };
    auto evaluate = [&](const std::vector<double>& xx, std::vector<double>* g, std::vector<double>* H) {
      std::vector<int> ids;
      return build(unpack(xx), exclude_registration, g, H, &ids);
    };
    auto plus = [](const std::vector<double>& xx, const std::vector<double>& d) {  // x, y, z plain; yaw: AngleLocalParameterization
      std::vector<double> y(xx.size());
      for (size_t i = 0; i < xx.size(); ++i) y[i] = (i % 4 == 3) ? normalizeAngle(xx[i] + d[i]) : xx[i] + d[i];
      return y;
    };
    Summary S = trustRegionMinimize(evaluate, plus, &x, solver_options);
    poses = unpack(x);
    return S;
  }

};

// The facade coxgraph's server calls (coxgraph/include/coxgraph/server/pose_graph_interface.h:19-108, the voxgraph base
// class behind it, and their call sites: src/server/coxgraph_server.cpp:452,466,500,509,517,523,544-550,
// src/server/visualizer/server_visualizer.cpp:28-56)
class PoseGraphInterface {
 public:
  enum class ConstraintType { RelPose, SubmapRelPose };  // printResiduals(ConstraintType), coxgraph_server.cpp:541-555
  typedef std::map<int, Pose4> PoseMap;

  // voxgraph's measurement_templates_.registration (config/server.yaml:28-36)
  struct RegistrationConfig {
    bool enabled = true;
    double sampling_ratio = 0.3;
    RegistrationPointType registration_point_type = RegistrationPointType::kIsosurfacePoints;  // "explicit_to_implicit"
    bool use_esdf_distance = true;
    double no_correspondence_cost = 0.0;
  };

  explicit PoseGraphInterface(SubmapCollection::Ptr submap_collection_ptr = SubmapCollection::Ptr()) : submap_collection_ptr_(std::move(submap_collection_ptr)) {
    // coxgraph/config/server.yaml:37-51
    const double lc[4] = {100.0, 100.0, 250.0, 250.0}, rp[4] = {1000.0, 1000.0, 2500.0, 2500.0};
    std::fill(lc_info_, lc_info_ + 16, 0.0);
    std::fill(sm_rp_info_, sm_rp_info_ + 16, 0.0);
    for (int i = 0; i < 4; ++i) {
      lc_info_[5 * i] = lc[i];
      sm_rp_info_[5 * i] = rp[i];
    }
  }
  // copyable: the final-mesh service optimises a what-if copy while the live graph keeps running (pose_graph_interface.h:52-62,
  // server_visualizer.cpp:28-31).  Constraints built from the collection get evaluators of their own (a handle has one
  // evaluation in flight); evaluators handed in by the caller stay shared.
  PoseGraphInterface(const PoseGraphInterface& rhs) { copyFrom(rhs, rhs.submap_collection_ptr_); }
  PoseGraphInterface(const PoseGraphInterface& rhs, SubmapCollection::Ptr submap_collection_ptr) { copyFrom(rhs, std::move(submap_collection_ptr)); }
  PoseGraphInterface& operator=(const PoseGraphInterface& rhs) {
    if (this != &rhs) copyFrom(rhs, rhs.submap_collection_ptr_);
    return *this;
  }

  void setVerbosity(bool verbose) { verbose_ = verbose; }
  void setRegistrationConfig(const RegistrationConfig& c) { registration_ = c; }
  const RegistrationConfig& getRegistrationConfig() const { return registration_; }
  void setLoopClosureInformation(const double info[16]) { std::copy(info, info + 16, lc_info_); }
  void setSubmapRelativePoseInformation(const double info[16]) { std::copy(info, info + 16, sm_rp_info_); }

  // pose_graph_interface.cpp:10-30 (non-robocentric): the node's initial pose is the submap's pose in the collection,
  // submap 0 is constant
  void addSubmap(SubmapID submap_id) {
    Transformation T;
    if (!submap_collection_ptr_ || !submap_collection_ptr_->getSubmapPose(submap_id, &T)) throw std::runtime_error("addSubmap: submap not in the collection");
    Pose4 p;
    pose4FromTransformation(T, p.v);
    clients_[static_cast<int>(submap_id)] = submap_collection_ptr_->getCliIdPairBySsid(submap_id).first;
    pose_graph_.addNode(static_cast<int>(submap_id), p, submap_id == 0);
  }
  // without a collection: the pose and the client are given
  void addSubmap(int submap_id, const Pose4& pose, int client_id = 0) {
    clients_[submap_id] = client_id;
    pose_graph_.addNode(submap_id, pose, submap_id == 0);
  }
  bool addLoopClosureMeasurement(int a, int b, const double T_ab[4]) {
    if (!pose_graph_.poses.count(a) || !pose_graph_.poses.count(b)) return false;
    pose_graph_.rel.emplace_back(a, b, T_ab, lc_info_);
    return true;
  }
  // addLoopClosureMeasurement(ser_sm_id_a, ser_sm_id_b, T_A_B, false)   (coxgraph_server.cpp:451-453)
  bool addLoopClosureMeasurement(SubmapID a, SubmapID b, const Transformation& T_A_B, bool /*publish*/ = false) {
    double t[4];
    pose4FromTransformation(T_A_B, t);
    return addLoopClosureMeasurement(static_cast<int>(a), static_cast<int>(b), t);
  }
  // pose_graph_interface.cpp:88-105: reference = first submap's registration points, reading = second submap's distance field
  void addForceRegistrationConstraint(SubmapID a, SubmapID b) { pose_graph_.reg.push_back(makeRegistrationConstraint(a, b)); }
  // with an evaluator of the caller's (not owned)
  void addForceRegistrationConstraint(int a, int b, RegistrationCostFunction* cost) {
    RegistrationConstraint c;
    c.a = a, c.b = b, c.cost = cost;
    pose_graph_.reg.push_back(c);
  }
  // voxgraph PoseGraphInterface::updateRegistrationConstraints (called at pose_graph_interface.cpp:38, between the two solves):
  // drop the overlap-driven constraints of the previous round, then constrain every pair of submaps whose surface boxes
  // overlap at the poses the first solve produced.  Forced constraints are a list of their own and stay.
  void updateRegistrationConstraints() {
    pose_graph_.resetRegistrationConstraints();
    overlapping_submap_list_.clear();
    if (!submap_collection_ptr_) return;
    updateSubmapCollectionPoses();
    const std::vector<SubmapID> ids = submap_collection_ptr_->getIDs();
    for (size_t i = 0; i < ids.size(); ++i) {
      if (!pose_graph_.poses.count(static_cast<int>(ids[i]))) continue;
      const VoxgraphSubmap::ConstPtr first = submap_collection_ptr_->getSubmapConstPtr(ids[i]);
      for (size_t j = i + 1; j < ids.size(); ++j) {
        if (!pose_graph_.poses.count(static_cast<int>(ids[j]))) continue;
        const VoxgraphSubmap::ConstPtr second = submap_collection_ptr_->getSubmapConstPtr(ids[j]);
        if (first->isFinished() && second->isFinished() && first->overlapsWith(*second)) overlapping_submap_list_.emplace_back(ids[i], ids[j]);
      }
    }
    for (const auto& pr : overlapping_submap_list_) pose_graph_.overlap_reg.push_back(makeRegistrationConstraint(pr.first, pr.second));
  }
  const std::vector<std::pair<SubmapID, SubmapID>>& getOverlappingSubmapList() const { return overlapping_submap_list_; }

  void resetSubmapRelativePoseConstrains() { pose_graph_.resetSubmapRelativePoseConstraints(); }  // (sic) pose_graph_interface.h:72-74
  // pose_graph_interface.cpp:51-71: consecutive submaps of one client keep their current relative pose
  void updateSubmapRPConstraints() {
    resetSubmapRelativePoseConstrains();
    std::map<int, std::vector<int>> by_client;
    if (submap_collection_ptr_) {
      for (int cid = 0; cid < submap_collection_ptr_->getClientNumber(); ++cid) {
        std::vector<SubmapID> ids;
        if (!submap_collection_ptr_->getSerSmIdsByCliId(cid, &ids)) continue;
        for (SubmapID s : ids)
          if (pose_graph_.poses.count(static_cast<int>(s))) by_client[cid].push_back(static_cast<int>(s));
      }
    } else {
      for (const auto& kv : clients_) by_client[kv.second].push_back(kv.first);  // std::map: ids ascending
    }
    for (const auto& kv : by_client)
      for (size_t k = 0; k + 1 < kv.second.size(); ++k) {
        const int i = kv.second[k], j = kv.second[k + 1];
        double T_ij[4];
        if (submap_collection_ptr_) {  // T_SMi_SMj = T_M_SMi^-1 * T_M_SMj from the collection's poses
          const Transformation Ti = submap_collection_ptr_->getSubmapPtr(static_cast<SubmapID>(i))->getPose(), Tj = submap_collection_ptr_->getSubmapPtr(static_cast<SubmapID>(j))->getPose();
          pose4FromTransformation(inverse(Ti) * Tj, T_ij);
        } else {
          const Pose4 &pa = pose_graph_.poses[i], &pb = pose_graph_.poses[j];
          const double c = std::cos(pa.v[3]), s = std::sin(pa.v[3]);
          const double d[3] = {pb.v[0] - pa.v[0], pb.v[1] - pa.v[1], pb.v[2] - pa.v[2]};
          T_ij[0] = c * d[0] + s * d[1], T_ij[1] = -s * d[0] + c * d[1], T_ij[2] = d[2], T_ij[3] = normalizeAngle(pb.v[3] - pa.v[3]);
        }
        addSubmapRelativePoseConstraint(i, j, T_ij);
      }
  }
  void addSubmapRelativePoseConstraint(int first_submap_id, int second_submap_id, const double T_S1_S2[4]) {
    pose_graph_.submap_rel.emplace_back(first_submap_id, second_submap_id, T_S1_S2, sm_rp_info_);
  }
  // The fork's check before a solve (coxgraph_server.cpp:509; implementation not in the reference tree): true when there
  // is at least one loop-closure candidate between two known submaps.  The server only logs the outcome.
  bool checkLoopClosureCandidates() const {
    for (const RelativePoseConstraint& c : pose_graph_.rel)
      if (pose_graph_.poses.count(c.a) && pose_graph_.poses.count(c.b)) return true;
    return false;
  }
  // pose_graph_interface.cpp:32-49: solve without registration constraints; refresh the overlap-driven registration
  // constraints at the loop-closed poses if enabled; solve with ALL constraints (forced registration constraints included,
  // whatever the flag says)
  std::pair<PoseGraph::Summary, PoseGraph::Summary> optimize(bool enable_registration = true) {
    const PoseGraph::Summary first = pose_graph_.optimize(true);
    if (enable_registration) updateRegistrationConstraints();
    const PoseGraph::Summary second = pose_graph_.optimize(false);
    return {first, second};
  }
  const PoseMap& getPoseMap() const { return pose_graph_.poses; }
  // the optimised node poses as transformations (yaw-only rotations: the 4-DoF pose graph drops roll and pitch)
  std::map<SubmapID, Transformation> getSubmapPoses() const {
    std::map<SubmapID, Transformation> out;
    for (const auto& kv : pose_graph_.poses) out[static_cast<SubmapID>(kv.first)] = transformationFromPose4(kv.second.v);
    return out;
  }
  // voxgraph PoseGraphInterface::updateSubmapCollectionPoses (server_visualizer.cpp:54): write the node poses back
  void updateSubmapCollectionPoses() {
    if (!submap_collection_ptr_) return;
    for (const auto& kv : pose_graph_.poses)
      if (submap_collection_ptr_->exists(static_cast<SubmapID>(kv.first))) submap_collection_ptr_->setSubmapPose(static_cast<SubmapID>(kv.first), transformationFromPose4(kv.second.v));
  }
  std::vector<double> evaluateResiduals(ConstraintType type) const {
    return pose_graph_.evaluateResiduals(type == ConstraintType::RelPose ? pose_graph_.rel : pose_graph_.submap_rel);
  }
  void printResiduals(ConstraintType type) const {  // pose_graph_interface.h:85-90
    for (double r : evaluateResiduals(type)) std::printf("%g ", r);
    std::printf("\n");
  }
  PoseGraph& poseGraph() { return pose_graph_; }
  const SubmapCollection::Ptr& submapCollection() const { return submap_collection_ptr_; }

 private:
  RegistrationConstraint makeRegistrationConstraint(SubmapID a, SubmapID b) const {
    if (!submap_collection_ptr_) throw std::runtime_error("registration constraint: no submap collection");
    const VoxgraphSubmap::ConstPtr first = submap_collection_ptr_->getSubmapConstPtr(a), second = submap_collection_ptr_->getSubmapConstPtr(b);
    if (!first || !second) throw std::runtime_error("registration constraint: unknown submap");
    const std::shared_ptr<RegistrationPointSet> pts = first->getRegistrationPoints(registration_.registration_point_type);
    RegistrationConstraint c;
    c.a = static_cast<int>(a), c.b = static_cast<int>(b);
    c.owned.reset(new RegistrationCostFunction(pts->handle(), second->getReadingLayer(registration_.use_esdf_distance), registration_.no_correspondence_cost,
                                               {std::shared_ptr<const void>(pts), std::shared_ptr<const void>(first), std::shared_ptr<const void>(second)}));
    c.cost = c.owned.get();
    c.sampling_ratio = registration_.sampling_ratio;
    return c;
  }
  void copyFrom(const PoseGraphInterface& rhs, SubmapCollection::Ptr collection) {
    pose_graph_ = rhs.pose_graph_;
    clients_ = rhs.clients_;
    std::copy(rhs.lc_info_, rhs.lc_info_ + 16, lc_info_);
    std::copy(rhs.sm_rp_info_, rhs.sm_rp_info_ + 16, sm_rp_info_);
    registration_ = rhs.registration_;
    overlapping_submap_list_ = rhs.overlapping_submap_list_;
    verbose_ = rhs.verbose_;
    submap_collection_ptr_ = std::move(collection);
    for (auto* list : {&pose_graph_.reg, &pose_graph_.overlap_reg})
      for (RegistrationConstraint& c : *list)
        if (c.owned) c = makeRegistrationConstraint(static_cast<SubmapID>(c.a), static_cast<SubmapID>(c.b));  // own evaluator, against the new collection's submaps
  }

  PoseGraph pose_graph_;
  SubmapCollection::Ptr submap_collection_ptr_;
  std::map<int, int> clients_;
  double lc_info_[16], sm_rp_info_[16];
  RegistrationConfig registration_;
  std::vector<std::pair<SubmapID, SubmapID>> overlapping_submap_list_;
  bool verbose_ = false;
};

}  // namespace coxgraph_hip
