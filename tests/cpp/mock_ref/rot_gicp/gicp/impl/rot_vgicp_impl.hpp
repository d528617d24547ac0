// DECLARATION-ONLY mock (see ../../../README.md): the template implementations live in the real headers only
#pragma once
