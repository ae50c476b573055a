#define YH_BUILD_ID "717889fabcd758ef"
