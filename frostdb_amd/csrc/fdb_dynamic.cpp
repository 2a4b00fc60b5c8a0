// fdb_dynamic.cpp — see fdb_dynamic.h.
#include "fdb_dynamic.h"

#include <cstring>
#include <unordered_map>

namespace fdb {

namespace {

const char* func_name(int32_t f) {
  switch (f) {
    case FDB_AGG_SUM: return "sum";
    case FDB_AGG_MIN: return "min";
    case FDB_AGG_MAX: return "max";
    case FDB_AGG_COUNT: return "count";
  }
  return "?";
}

bool has_prefix(const std::string& field, const std::string& prefix) {  // DynamicColumn.MatchColumn (logicalplan/expr.go:564-566)
  return field.size() > prefix.size() && field.compare(0, prefix.size(), prefix) == 0 && field[prefix.size()] == '.';
}

// Reads group-key columns of a finished result (any mix of own / external buffers) for the join.
struct KeyReader {
  const OutColumn* c;
  const uint8_t* vals;
  const uint8_t* bits;
  explicit KeyReader(const OutColumn& col) : c(&col) {
    vals = col.ext_values != nullptr ? col.ext_values : col.values.data();
    bits = col.null_count > 0 ? (col.ext_validity != nullptr ? col.ext_validity : (col.validity.empty() ? nullptr : col.validity.data())) : nullptr;
  }
  bool valid(int64_t i) const { return bits == nullptr || ((bits[i >> 3] >> (i & 7)) & 1); }
  // Appends one self-delimiting field: NULL and (for integer keys) 0 are the same group (hashed.go:254-272), so both
  // serialise as the value 0.
  void append(int64_t i, std::string* k) const {
    const bool ok = valid(i);
    if (c->is_dict) {
      if (!ok) { k->push_back('\0'); return; }
      uint32_t idx; std::memcpy(&idx, vals + (size_t)i * 4, 4);
      const int64_t b0 = c->dict_offset((int64_t)idx), b1 = c->dict_offset((int64_t)idx + 1);
      put_bytes(c->dict_bytes() + b0, (size_t)(b1 - b0), k);
    } else if (c->is_str) {
      if (!ok) { k->push_back('\0'); return; }
      const bool wide = c->format == "U" || c->format == "Z";
      int64_t b0, b1;
      if (wide) { std::memcpy(&b0, vals + (size_t)i * 8, 8); std::memcpy(&b1, vals + (size_t)(i + 1) * 8, 8); }
      else { int32_t a, b; std::memcpy(&a, vals + (size_t)i * 4, 4); std::memcpy(&b, vals + (size_t)(i + 1) * 4, 4); b0 = a; b1 = b; }
      put_bytes(c->str_data.data() + b0, (size_t)(b1 - b0), k);
    } else if (c->format == "b") {
      k->push_back(!ok ? '\0' : ((vals[i >> 3] >> (i & 7)) & 1) ? '\2' : '\1');
    } else {
      unsigned long long v = 0;
      if (ok) std::memcpy(&v, vals + (size_t)i * 8, 8);
      k->push_back('\1');
      k->append((const char*)&v, 8);
    }
  }
  static void put_bytes(const char* p, size_t n, std::string* k) {
    k->push_back('\1');
    const uint64_t len = n;
    k->append((const char*)&len, 8);
    k->append(p, n);
  }
};

}  // namespace

// ---- DescCopy ------------------------------------------------------------------------------------------------------------
DescCopy::DescCopy(const fdb_plan_desc* d) {
  check_desc_shape(d);
  auto lit = [&](fdb_literal l) {
    if (l.data != nullptr && l.len > 0) { strs_.emplace_back(l.data, (size_t)l.len); l.data = strs_.back().data(); }
    else { l.data = nullptr; l.len = 0; }
    return l;
  };
  for (int32_t i = 0; i < d->n_filter; i++) {
    fdb_expr e = d->filter[i];
    if (e.column != nullptr) e.column = keep(e.column);
    e.literal = lit(e.literal);
    filter_.push_back(e);
  }
  filter_root_ = d->filter_root;
  for (int32_t i = 0; i < d->n_aggs; i++) {
    fdb_aggregation a = d->aggs[i];
    if (a.column == nullptr) throw Error(FDB_ERR_INVALID, "aggregation without a column");
    a.column = keep(a.column);
    if (a.dynamic != 0) {
      if (a.func != FDB_AGG_SUM && a.func != FDB_AGG_MIN && a.func != FDB_AGG_MAX && a.func != FDB_AGG_COUNT)
        throw Error(FDB_ERR_UNSUPPORTED, std::string("aggregation function over the dynamic column set ") + a.column + ".*: only sum, min, max and count");
      dynamic_aggs.push_back(a);
    } else {
      static_aggs.push_back(a);
    }
  }
  for (int32_t i = 0; i < d->n_groups; i++) {
    fdb_group_expr g = d->groups[i];
    if (g.name != nullptr) g.name = keep(g.name);
    groups_.push_back(g);
  }
  n_groups = d->n_groups;
  final_stage = d->final_stage != 0;
  proj_nodes_.resize((size_t)d->n_projections);
  for (int32_t i = 0; i < d->n_projections; i++) {
    fdb_projection p = d->projections[i];
    if (p.name != nullptr) p.name = keep(p.name);
    for (int32_t k = 0; k < p.n_nodes; k++) {
      fdb_proj_node n = p.nodes[k];
      if (n.column != nullptr) n.column = keep(n.column);
      n.literal = lit(n.literal);
      proj_nodes_[(size_t)i].push_back(n);
    }
    p.nodes = proj_nodes_[(size_t)i].data();
    projs_.push_back(p);
  }
  re_fn_ = d->regex_match;
  re_user_ = d->regex_user;
}

fdb_plan_desc DescCopy::view(const std::vector<fdb_aggregation>& aggs, bool final) const {
  fdb_plan_desc v;
  std::memset(&v, 0, sizeof(v));
  v.filter = filter_.empty() ? nullptr : filter_.data();
  v.n_filter = (int32_t)filter_.size();
  v.filter_root = filter_root_;
  v.aggs = aggs.empty() ? nullptr : aggs.data();
  v.n_aggs = (int32_t)aggs.size();
  v.groups = groups_.empty() ? nullptr : groups_.data();
  v.n_groups = (int32_t)groups_.size();
  v.final_stage = final ? 1 : 0;
  v.projections = projs_.empty() ? nullptr : projs_.data();
  v.n_projections = (int32_t)projs_.size();
  v.regex_match = re_fn_;
  v.regex_user = re_user_;
  return v;
}

// ---- DynamicAggs -----------------------------------------------------------------------------------------------------------
bool DynamicAggs::wanted(const fdb_plan_desc* d) {
  if (d == nullptr || d->n_aggs <= 0 || d->aggs == nullptr) return false;  // (a malformed descriptor is refused by whoever reads it next)
  for (int32_t i = 0; i < d->n_aggs; i++) if (d->aggs[i].dynamic != 0) return true;
  return false;
}

DynamicAggs::DynamicAggs(const fdb_plan_desc* d, int device) : desc_(d), device_(device) {}

DynamicAggs::Child* DynamicAggs::child_for(const std::string& field, int32_t func, bool create) {
  for (auto& c : children_) if (c->column == field && c->func == func) return c.get();
  if (!create) return nullptr;
  auto c = std::make_unique<Child>();
  c->column = field;
  c->func = func;
  // the partial stage hands the field on under its own name; the final stage names it func(field) (aggregate.go:313-334)
  c->result_name = desc_.final_stage ? std::string(func_name(func)) + "(" + field + ")" : field;
  // One ordinary aggregation over the concrete column. A final-stage family reads partial results (columns named after the
  // field): partial counts are summed (runAggregation, aggregate.go:965-969), the other functions merge with themselves —
  // which is exactly a NON-final plan with that function over that column.
  fdb_aggregation a;
  std::memset(&a, 0, sizeof(a));
  a.func = (desc_.final_stage && func == FDB_AGG_COUNT) ? FDB_AGG_SUM : func;
  a.column = desc_.keep(field);
  std::vector<fdb_aggregation> aggs{a};
  const fdb_plan_desc v = desc_.view(aggs, /*final_stage=*/false);
  c->plan = std::make_unique<Plan>(&v, device_);
  children_.push_back(std::move(c));
  return children_.back().get();
}

void DynamicAggs::match(const std::vector<std::string>& fields, std::vector<Child*>* hit) {
  hit->clear();
  for (const std::string& f : fields)
    for (const fdb_aggregation& d : desc_.dynamic_aggs)
      if (has_prefix(f, d.column)) hit->push_back(child_for(f, d.func, true));
  if (hit->empty())  // aggregate.go:366-380: at least one dynamic column per record
    throw Error(FDB_ERR_NOT_FOUND, "aggregate field(s) not found, aggregations are not possible without it (no column of the dynamic aggregation in this record)");
}

void DynamicAggs::push(Plan& main, const ArrowArray* array, const ArrowSchema* schema) {
  HostRecordView view;
  view_record(array, schema, &view);
  std::vector<std::string> fields;
  for (const HostColView& c : view.cols) fields.push_back(c.name);
  std::vector<Child*> hit;
  match(fields, &hit);
  if (main_active()) main.push(array, schema);
  for (Child* c : hit) c->plan->push(array, schema);
}

void DynamicAggs::push_batches(Plan& main, const DeviceBatch* const* bs, int n) {
  std::vector<std::vector<const DeviceBatch*>> per_child;
  std::vector<Child*> order;
  for (int i = 0; i < n; i++) {
    std::vector<std::string> fields;
    for (const DevColumn& c : bs[i]->cols) fields.push_back(c.name);
    std::vector<Child*> hit;
    match(fields, &hit);
    for (Child* c : hit) {
      size_t k = 0;
      for (; k < order.size(); k++) if (order[k] == c) break;
      if (k == order.size()) { order.push_back(c); per_child.emplace_back(); }
      per_child[k].push_back(bs[i]);
    }
  }
  if (main_active()) { main.settle(); main.push_batches(bs, n); }
  for (size_t k = 0; k < order.size(); k++) { order[k]->plan->settle(); order[k]->plan->push_batches(per_child[k].data(), (int)per_child[k].size()); }
}

const char* DynamicAggs::draw(Plan& main) {
  std::string s = main.draw();
  if (desc_.static_aggs.empty()) {  // main is only the group keys (drawn as a Distinction) or nothing at all
    const std::string suffix = " [gfx950]";
    std::string body = s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0 ? s.substr(0, s.size() - suffix.size()) : s;
    const size_t at = body.find("Distinction (");
    if (at != std::string::npos) body.replace(at, 13, "HashAggregate ( by ");
    else body += std::string(body.empty() ? "" : " - ") + "HashAggregate ( by )";
    s = body + suffix;
  }
  draw_ = s;
  return draw_.c_str();
}

void DynamicAggs::settle(Plan& main) {
  main.settle();
  for (auto& c : children_) c->plan->settle();
}

int64_t DynamicAggs::num_groups(Plan& main) {
  settle(main);
  if (main_active()) return main.num_groups();
  for (auto& c : children_) if (c->plan->num_groups() > 0) return 1;
  return 0;
}

void DynamicAggs::merge_from(Plan& main, DynamicAggs& src, Plan& src_main) {
  if (src.desc_.dynamic_aggs.size() != desc_.dynamic_aggs.size() || src.main_active() != main_active())
    throw Error(FDB_ERR_INVALID, "plans have different aggregations");
  settle(main);
  src.settle(src_main);
  if (main_active()) main.merge_from(src_main);
  for (auto& sc : src.children_) child_for(sc->column, sc->func, true)->plan->merge_from(*sc->plan);
}

void DynamicAggs::finish(Plan& main, ArrowArray* out, ArrowSchema* out_schema, int64_t* n_rows) {
  settle(main);
  std::vector<OutColumn> cols;
  int64_t n = 0;
  size_t n_keys = 0;
  if (main_active()) {
    n = main.finish_columns(&cols);
    n_keys = main.n_key_columns();
  }
  // children's results
  struct Res { Child* c; std::vector<OutColumn> cols; int64_t n; size_t n_keys; };
  std::vector<Res> res;
  for (auto& c : children_) {
    Res r;
    r.c = c.get();
    r.n = c->plan->finish_columns(&r.cols);
    r.n_keys = c->plan->n_key_columns();
    if (!main_active() && r.n > 0) n = 1;  // no grouping, no static aggregation: one row as soon as any row was aggregated
    res.push_back(std::move(r));
  }
  cols.reserve(cols.size() + res.size());  // (the readers below point at elements of `cols`)
  // main's rows by key tuple
  std::unordered_map<std::string, int64_t> row_of;
  std::vector<KeyReader> main_keys;
  for (size_t k = 0; k < n_keys; k++) main_keys.emplace_back(cols[k]);
  if (n_keys > 0) {
    row_of.reserve((size_t)n * 2);
    std::string key;
    for (int64_t i = 0; i < n; i++) {
      key.clear();
      for (const KeyReader& kr : main_keys) kr.append(i, &key);
      row_of.emplace(key, i);
    }
  }
  for (Res& r : res) {
    // A column whose records never had a selected row never reached the reference's HashAggregate (PredicateFilter drops empty
    // records, filter.go:264-266), so it was never converted into an aggregation: no result column.
    if (r.n == 0) continue;
    const OutColumn& src = r.cols.back();  // the child's one aggregation column
    OutColumn o;
    o.name = r.c->result_name;
    o.format = src.format;
    o.length = n;
    o.values.assign((size_t)n * 8, 0);
    // a group the child never saw: SUM / COUNT of nothing is 0, MIN / MAX of an empty array is NULL (aggregate.go:806-809, :884-887)
    const bool null_when_absent = r.c->func == FDB_AGG_MIN || r.c->func == FDB_AGG_MAX;
    std::vector<uint8_t> seen((size_t)n, 0);
    const uint8_t* sv = src.ext_values != nullptr ? src.ext_values : src.values.data();
    // the child's key columns in main's column order (a column the child never saw is NULL there)
    std::vector<int> child_col(n_keys, -1);
    for (size_t k = 0; k < n_keys; k++)
      for (size_t j = 0; j < r.n_keys; j++) if (r.cols[j].name == cols[k].name) child_col[k] = (int)j;
    std::vector<KeyReader> child_keys;
    for (size_t j = 0; j < r.n_keys; j++) child_keys.emplace_back(r.cols[j]);
    std::string key;
    for (int64_t i = 0; i < r.n; i++) {
      int64_t row = 0;
      if (n_keys > 0) {
        key.clear();
        for (size_t k = 0; k < n_keys; k++) {
          if (child_col[k] < 0) {  // absent column ≡ NULL: the same encoding KeyReader::append gives a NULL of main's column type
            const OutColumn& mc = cols[k];
            if (mc.is_dict || mc.is_str || mc.format == "b") key.push_back('\0');
            else { key.push_back('\1'); key.append(8, '\0'); }
          } else {
            child_keys[(size_t)child_col[k]].append(i, &key);
          }
        }
        auto it = row_of.find(key);
        if (it == row_of.end()) throw Error(FDB_ERR_INVALID, "internal: a dynamic aggregation produced a group the main plan does not have");
        row = it->second;
      }
      std::memcpy(o.values.data() + (size_t)row * 8, sv + (size_t)i * 8, 8);
      seen[(size_t)row] = 1;
    }
    if (null_when_absent) {
      o.validity.assign((size_t)(n + 7) / 8, 0);
      for (int64_t i = 0; i < n; i++) {
        if (seen[(size_t)i]) o.validity[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
        else o.null_count++;
      }
      if (o.null_count == 0) o.validity.clear();
    }
    cols.push_back(std::move(o));
  }
  if (n_rows) *n_rows = n;
  export_record(std::move(cols), n, out, out_schema);
}

}  // namespace fdb
