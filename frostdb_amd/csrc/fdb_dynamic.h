// fdb_dynamic.h — aggregations over a DynamicColumn (`max(foo)` over every `foo.*` column).
//
// Reference: Aggregate() marks an aggregation whose expression contains a DynamicColumn (aggregate.go:38-46),
// NewHashAggregate keeps those apart (:168-176) and HashAggregate.Callback turns each matching FIELD into a concrete
// aggregation the first time a record carries it (:306-336): result column named after the field in the partial stage,
// `max(foo.bar)` in the final stage (resultNameWithConcreteColumn, :973-990); a record that lacks the field contributes
// nothing to it (:471-475); a record that matches none of the dynamic aggregations is an error (:366-380). Pinned by
// Test_Aggregation_DynCol (root aggregate_test.go:436-519: three columns that come and go, no grouping).
//
// Here: the set of concrete columns is not known when the plan is built and grows while records arrive, but every concrete
// aggregation is an ordinary one-aggregation plan over the same filter and the same group matchers. So a plan with dynamic
// aggregations is a FAMILY of ordinary plans — `main` (the static aggregations, or only the group keys) and one child per
// concrete column, created when the column is first seen — each record goes to main and to the children whose column it
// carries, and Finish joins the children's results to main's rows by group-key tuple. Nothing in the kernels knows about it.
// The price is one scan per child over the filter and group columns; dynamic aggregations are rare and mostly ungrouped
// (the reference itself only survives them without grouping: a record that lacks an aggregated column and creates a new
// group dereferences nil, :413-417).
#pragma once

#include <deque>
#include <memory>
#include <string>
#include <vector>

#include "fdb_plan.h"

namespace fdb {

// Deep copy of a descriptor (the caller's strings only live for the duration of fdb_plan_create).
class DescCopy {
 public:
  explicit DescCopy(const fdb_plan_desc* d);
  DescCopy(const DescCopy&) = delete;             // (the copied nodes point into this object's own strings)
  DescCopy& operator=(const DescCopy&) = delete;
  // The descriptor again with `aggs` as its aggregation list (pointers stay valid as long as *this and `aggs` do).
  fdb_plan_desc view(const std::vector<fdb_aggregation>& aggs, bool final_stage) const;
  const char* keep(const std::string& s) { strs_.push_back(s); return strs_.back().c_str(); }
  std::vector<fdb_aggregation> static_aggs;   // dynamic == 0
  std::vector<fdb_aggregation> dynamic_aggs;  // dynamic != 0: `column` is the prefix
  bool final_stage = false;
  int32_t n_groups = 0;

 private:
  std::deque<std::string> strs_;
  std::vector<fdb_expr> filter_;
  std::vector<fdb_group_expr> groups_;
  std::vector<std::vector<fdb_proj_node>> proj_nodes_;
  std::vector<fdb_projection> projs_;
  int32_t filter_root_ = -1;
  fdb_regex_match_fn re_fn_ = nullptr;
  void* re_user_ = nullptr;
};

class DynamicAggs {
 public:
  DynamicAggs(const fdb_plan_desc* d, int device);
  static bool wanted(const fdb_plan_desc* d);                  // the descriptor has a dynamic aggregation
  fdb_plan_desc main_desc() const { return desc_.view(desc_.static_aggs, desc_.final_stage); }
  bool main_active() const { return !desc_.static_aggs.empty() || desc_.n_groups > 0; }

  void push(Plan& main, const ArrowArray* array, const ArrowSchema* schema);
  void push_batches(Plan& main, const DeviceBatch* const* bs, int n);
  void finish(Plan& main, ArrowArray* out, ArrowSchema* out_schema, int64_t* n_rows);
  void merge_from(Plan& main, DynamicAggs& src, Plan& src_main);
  int64_t num_groups(Plan& main);
  void settle(Plan& main);
  // Draw: like HashAggregate.Draw (aggregate.go:226-243) only the aggregations that exist so far are listed — the static ones.
  const char* draw(Plan& main);

 private:
  struct Child {
    std::string column;       // concrete field name
    int32_t func = 0;         // the dynamic aggregation's function (names the result)
    std::string result_name;  // the field name (partial stage) or func(field) (final stage)
    std::unique_ptr<Plan> plan;
  };
  Child* child_for(const std::string& field, int32_t func, bool create);
  void match(const std::vector<std::string>& fields, std::vector<Child*>* hit);  // children of the fields a record carries (created on first sight)

  DescCopy desc_;
  int device_;
  std::vector<std::unique_ptr<Child>> children_;  // creation order = output order
  std::string draw_;
};

}  // namespace fdb
