#include "fdb_arrow.h"
#include <string>
static thread_local std::string g_err;
extern "C" const char* fdb_last_error(void) { return g_err.c_str(); }
extern "C" int fdb_arrow_roundtrip(struct ArrowArray* batch, struct ArrowSchema* schema, struct ArrowArray* out, struct ArrowSchema* out_schema) {
  try {
    fdb::HostRecordView view;
    fdb::view_record(batch, schema, &view);
    fdb::roundtrip_record(view, out, out_schema);
    return 0;
  } catch (const fdb::Error& e) { g_err = e.what(); return e.code; }
}
