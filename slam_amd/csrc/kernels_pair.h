#pragma once
#include "rvo_dev.h"
struct StreamState { int dummy; };
