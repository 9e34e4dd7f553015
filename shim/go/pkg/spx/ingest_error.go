package spx

/*
#include "spx.h"
*/
import "C"

import "errors"

func fmtIngestError(h *C.spx_ingest) error { return errors.New(C.GoString(C.spx_ingest_error(h))) }
