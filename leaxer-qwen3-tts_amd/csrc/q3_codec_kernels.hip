// placeholder — replaced by the codec decoder kernels
#include "q3_common.h"
