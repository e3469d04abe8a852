// placeholder main until the drop-in CLI lands (next commit)
#include <cstdio>
#include "../../include/sipnet_amd.h"
int main() { std::printf("%s\n", sipnet_version()); return 0; }
