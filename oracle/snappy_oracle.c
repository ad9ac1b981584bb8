#include "oracle.h"
