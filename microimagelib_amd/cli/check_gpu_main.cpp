// checkGPUDevice -- list the GPUs the library can see (reference src/check_gpu.cpp: a call to queryDevice()).
#include "../../include/libapi.h"

int main() {
    queryDevice();
    return 0;
}
