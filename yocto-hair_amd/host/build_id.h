#define YH_BUILD_ID "8f51eb1a7e0ff2bf"
