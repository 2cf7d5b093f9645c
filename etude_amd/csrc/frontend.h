#pragma once
#include "common.h"
#include "../../include/etude_hip.h"
