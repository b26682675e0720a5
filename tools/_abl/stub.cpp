// error-sink stub so that wino_fused.hip links on its own for the ablation builds
#include <stdarg.h>
void cslam_set_error(const char *fmt, ...) { (void)fmt; }
