#include "ATen.h"
