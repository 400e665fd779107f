// placeholder, replaced by the acoustic model
#include "qv_common.h"
int qv_model_create(qv_engine *eng, const qv_config *, QvModel **) { qv_set_error(eng, "model not built"); return QV_ERR_NO_MODEL; }
void qv_model_destroy(QvModel *) {}
int qv_model_forward(qv_engine *eng, QvModel *, const float *, const int64_t *, int, int64_t, float *, int, int32_t *, hipStream_t) { return QV_ERR_NO_MODEL; }
int qv_model_tap(qv_engine *eng, QvModel *, int, int, float *, hipStream_t) { return QV_ERR_NO_MODEL; }
