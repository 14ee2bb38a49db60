// temporary stubs until the extraction / LC / pose-graph translation units land
#include "dsss_internal.h"
void dsss_pg_free(dsss_ctx*) {}
