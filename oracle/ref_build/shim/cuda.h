#pragma once
#define CUDA_VERSION 11080
#include "cuda_runtime.h"
